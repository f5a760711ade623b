"""Deformable-convolution sampling, rotated NMS, fused anchor targets + detection losses (csrc/dcn_sample.hip, nms_rotated.hip, anchor_loss.hip).
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
from ._core import _on, _ptr, _raw_stream, _same_device, _stream, _want, _workspace



# ---------------------------------------------------------------------------------------------
# deformable 3x3 sampling (DepthNet's DCN)
# ---------------------------------------------------------------------------------------------
class _DcnSample(torch.autograd.Function):
    """x (B,H,W,C) bf16 or fp32, offset (B,Ho,Wo,18) fp32 -> col (B*Ho*Wo, 9*C) in x's type (row gathers, no atomics)."""

    @staticmethod
    def forward(ctx, x, offset, stride, pad, dil):
        B, H, W, C = x.shape
        Ho, Wo = offset.shape[1:3]
        col = torch.empty((B * Ho * Wo, 9 * C), dtype=x.dtype, device=x.device)
        sfx = "_f32" if x.dtype == torch.float32 else ""
        with _on(x.device):
            check(getattr(lib(), "omnihd_dcn3x3_sample_fwd" + sfx)(_ptr(x), _ptr(offset), _ptr(col), B, H, W, C, stride, pad,
                                                                   dil, _stream()), "omnihd_dcn3x3_sample_fwd" + sfx)
        ctx.save_for_backward(x, offset)
        ctx.geo = (stride, pad, dil)
        return col

    @staticmethod
    def backward(ctx, gcol):
        x, offset = ctx.saved_tensors
        stride, pad, dil = ctx.geo
        B, H, W, C = x.shape
        gcol = gcol.contiguous().to(x.dtype)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        goff = torch.empty_like(offset) if ctx.needs_input_grad[1] else None
        radius = offset.abs().amax().ceil().to(torch.int32).reshape(1)          # stays on the device
        sfx = "_f32" if x.dtype == torch.float32 else ""
        with _on(x.device):
            check(getattr(lib(), "omnihd_dcn3x3_sample_bwd" + sfx)(_ptr(x), _ptr(offset), _ptr(gcol), _ptr(radius), _ptr(gx),
                                                                   _ptr(goff), B, H, W, C, stride, pad, dil, _stream()),
                  "omnihd_dcn3x3_sample_bwd" + sfx)
        return gx, goff, None, None, None


def dcn3x3_sample(x_nhwc, offset_nhwc, stride=1, pad=1, dil=1):
    if not (x_nhwc.is_cuda and x_nhwc.dtype in (torch.bfloat16, torch.float32) and x_nhwc.is_contiguous()):
        raise TypeError("x must be a contiguous (B,H,W,C) bf16 or fp32 CUDA(HIP) tensor")
    if not (offset_nhwc.dtype == torch.float32 and offset_nhwc.is_contiguous() and offset_nhwc.shape[-1] == 18):
        raise TypeError("offset must be a contiguous (B,Ho,Wo,18) fp32 tensor")
    return _DcnSample.apply(x_nhwc, offset_nhwc, int(stride), int(pad), int(dil))


def dcn3x3_supported(x, k, stride, deform_groups):
    return (x.is_cuda and x.dim() == 4 and k == 3 and stride == 1 and deform_groups == 1 and x.shape[1] in (32, 64, 128, 256))


# --------------------------------------------------------------------------------------------
# Test-time post-process: rotated BEV NMS (mmdet3d v0.17.1 `nms_gpu` / `boxes_iou_bev`)
# --------------------------------------------------------------------------------------------
def nms_rotated(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """mmdet3d `nms_gpu(boxes, scores, thresh, pre_maxsize, post_max_size)`: boxes (N,5) fp32
    (x1, y1, x2, y2, ry), scores (N,) -> int64 indices of the kept boxes in descending score
    order.  Sorting stays on torch (as upstream); masks and their reduction run in the HIP library."""
    _want(boxes, torch.float32, "boxes")
    _same_device(boxes, scores)
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    sorted_boxes = boxes[order].contiguous()
    n = sorted_boxes.shape[0]
    dev = boxes.device
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    num_out = torch.zeros(1, dtype=torch.int32, device=dev)
    with _on(dev):
        ws = _workspace(lib().omnihd_nms_rotated_workspace_bytes(n), dev)
        check(lib().omnihd_nms_rotated(_ptr(sorted_boxes), n, float(thresh), _ptr(keep), _ptr(num_out), _ptr(ws),
                                       ws.numel(), _stream()), "omnihd_nms_rotated")
    kept = order[keep[:int(num_out.item())]].contiguous()
    if post_max_size is not None:
        kept = kept[:post_max_size]
    return kept


def iou_bev_matrix(boxes_a, boxes_b):
    """(Na,5) x (Nb,5) (x1,y1,x2,y2,ry) -> (Na,Nb) rotated BEV IoU (mmdet3d `boxes_iou_bev`)."""
    _want(boxes_a, torch.float32, "boxes_a")
    _want(boxes_b, torch.float32, "boxes_b")
    _same_device(boxes_a, boxes_b)
    out = torch.empty((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    with _on(boxes_a.device):
        check(lib().omnihd_iou_bev_matrix(_ptr(boxes_a), boxes_a.shape[0], _ptr(boxes_b), boxes_b.shape[0],
                                          _ptr(out), _stream()), "omnihd_iou_bev_matrix")
    return out


# --------------------------------------------------------------------------------------------
# Anchor target assignment + detection losses, fused (csrc/anchor_loss.hip)
# --------------------------------------------------------------------------------------------
class _AnchorLoss(torch.autograd.Function):
    """(cls_score, bbox_pred, dir_pred) -> (loss_cls, loss_bbox, loss_dir) of Anchor3DHead.loss for one feature level: target
    assignment, the three losses and the (unscaled) gradients of the three maps in three launches; the backward scales the saved
    gradient maps in one.  See include/omnihd_hip.h: omnihd_anchor_loss_fwd."""

    @staticmethod
    def forward(ctx, cls_score, bbox_pred, dir_pred, anchors, gt_boxes, gt_labels, gt_offsets, meta):
        (num_classes, code_size, na, params7, sin_diff, code_weight, loss_weights) = meta
        # (fp32 and dense in NCHW or NHWC memory: ``anchor_loss`` casts / copies OUTSIDE the function where needed)
        B, _, H, W = cls_score.shape
        dev = cls_score.device
        total_gt = int(gt_boxes.shape[0])
        g_cls, g_box, g_dir = torch.empty_like(cls_score), torch.empty_like(bbox_pred), torch.empty_like(dir_pred)
        assert g_cls.stride() == cls_score.stride() and g_box.stride() == bbox_pred.stride() and g_dir.stride() == dir_pred.stride()
        out = torch.empty(4 + B, dtype=torch.float32, device=dev)
        strides = (ctypes.c_longlong * 12)(*cls_score.stride(), *bbox_pred.stride(), *dir_pred.stride())
        h_par = (ctypes.c_float * 7)(*[float(v) for v in params7])
        h_cw = (ctypes.c_float * code_size)(*[float(v) for v in code_weight])
        h_lw = (ctypes.c_float * 3)(*[float(v) for v in loss_weights])
        L = lib()
        with _on(dev):
            ws = _workspace(L.omnihd_anchor_loss_workspace_bytes(B, H * W * na, total_gt), dev)
            check(L.omnihd_anchor_loss_fwd(_ptr(anchors), _ptr(gt_boxes) if total_gt else None, _ptr(gt_labels) if total_gt else None,
                                           _ptr(gt_offsets), total_gt, _ptr(cls_score), _ptr(bbox_pred), _ptr(dir_pred), B, H, W, na,
                                           num_classes, code_size, ctypes.cast(strides, ctypes.c_void_p), ctypes.cast(h_par, ctypes.c_void_p),
                                           1 if sin_diff else 0, ctypes.cast(h_cw, ctypes.c_void_p), ctypes.cast(h_lw, ctypes.c_void_p),
                                           _ptr(g_cls), _ptr(g_box), _ptr(g_dir), _ptr(out), _ptr(ws), ws.numel(), _raw_stream()),
                  "omnihd_anchor_loss_fwd")
        ctx.save_for_backward(g_cls, g_box, g_dir, out)
        ctx.loss_weights = tuple(float(v) for v in loss_weights)
        info = out[3:]
        ctx.mark_non_differentiable(info)
        return out[0], out[1], out[2], info

    @staticmethod
    def backward(ctx, up_cls, up_box, up_dir, _up_info):
        if getattr(ctx, "consumed", False):
            # the gradient maps are scaled IN PLACE below (raw pointers: autograd's version counters do not see it): a second
            # backward over the same graph would scale them twice and return wrong gradients without an error
            raise RuntimeError("omnihd anchor loss: a second backward pass over the same forward is not supported (the saved "
                               "gradient maps are consumed in place); run the forward again, or set OMNIHD_ANCHOR_LOSS=0")
        ctx.consumed = True
        g_cls, g_box, g_dir, out = ctx.saved_tensors
        dev = g_cls.device
        h_lw = (ctypes.c_float * 3)(*ctx.loss_weights)
        ups = [None if u is None else u.to(torch.float32).contiguous() for u in (up_cls, up_box, up_dir)]
        with _on(dev):
            check(lib().omnihd_anchor_loss_bwd(_ptr(g_cls), g_cls.numel(), _ptr(g_box), g_box.numel(), _ptr(g_dir), g_dir.numel(),
                                               None if ups[0] is None else _ptr(ups[0]), None if ups[1] is None else _ptr(ups[1]),
                                               None if ups[2] is None else _ptr(ups[2]), _ptr(out), ctypes.cast(h_lw, ctypes.c_void_p),
                                               _raw_stream()), "omnihd_anchor_loss_bwd")
        return g_cls, g_box, g_dir, None, None, None, None, None


def anchor_loss(cls_score, bbox_pred, dir_pred, anchors, gt_boxes, gt_labels, gt_offsets, num_classes, code_size, anchors_per_loc,
                pos_iou_thr, neg_iou_thr, min_pos_iou, gamma, alpha, beta, dir_offset, sin_diff, code_weight, loss_weights):
    """Fused Anchor3DHead loss (one feature level).  gt_boxes (total, code_size) fp32 / gt_labels (total,) int32 concatenated over
    the batch, gt_offsets (B+1,) int32 on the device.  Returns (loss_cls, loss_bbox, loss_dir, info) with info = [avg_factor,
    positives per sample...]."""
    meta = (int(num_classes), int(code_size), int(anchors_per_loc),
            (pos_iou_thr, neg_iou_thr, min_pos_iou, gamma, alpha, beta, dir_offset), bool(sin_diff), tuple(code_weight), tuple(loss_weights))
    dense = lambda t: t if (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)) else t.contiguous()
    maps = [dense(t.float()) for t in (cls_score, bbox_pred, dir_pred)]          # differentiable casts: bf16 maps get bf16 gradients
    return _AnchorLoss.apply(maps[0], maps[1], maps[2], anchors, gt_boxes, gt_labels, gt_offsets, meta)
