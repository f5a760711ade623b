"""BatchNorm epilogues as autograd Functions: frozen (affine + residual + ReLU) and training mode (csrc/affine_act.hip, batch_norm.hip).
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
from ._core import FAST_PATHS, _SIZE_CACHE, _f32c, _on, _raw_stream, _rows_view, _want, _wgrad_workspace
from .policy import _fp32_policy, f16_handover
from .planes import _alloc_planes, _amax_slot, half_wanted, planes_wanted, tag_half, tag_planes, tag_producer



# --------------------------------------------------------------------------------------------
# Frozen-BatchNorm epilogue: y = act(x * scale + shift (+ residual)), channels-last bf16
# --------------------------------------------------------------------------------------------
class _AffineAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, res, relu):
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        # fp32 outputs are handed to the convolution that reads them next as operand planes once it has asked (see take_planes /
        # take_half): the hi / lo bf16 planes of the fp32-grade form, or the half plane of the TF32-grade form (OMNIHD_FP32_CONV=f16)
        # (the hi / lo hand-over from THIS epilogue is opt-in, OMNIHD_SPLIT_HANDOVER=all: bit-identical by test, 45.00 / 44.72 -> 44.62 / 44.63 ms
        # in the step (gpurun r6_31) — the GPU suite has not run with it as the default)
        f32 = x.dtype == torch.float32
        f16 = f32 and _fp32_policy() == "f16"
        split_all = f32 and not f16 and _env("OMNIHD_SPLIT_HANDOVER", "1") == "all"
        pkey = ("aff_y", scale.data_ptr()) if (f16 or split_all) else None
        y16 = torch.empty_like(x, dtype=torch.float16) if (f16 and half_wanted(pkey)) else None
        y_planes = _alloc_planes(y) if (split_all and planes_wanted(pkey)) else None
        with _on(x.device):
            if y16 is not None or y_planes is not None:
                hi, lo = (y16.data_ptr(), None) if y16 is not None else (y_planes[0].data_ptr(), y_planes[1].data_ptr())
                check(lib().omnihd_affine_act_fwd_f32_planes(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), None if res is None else res.data_ptr(),
                                                             y.data_ptr(), hi, lo, n * h * w, c, 1 if relu else 0, _raw_stream()),
                      "omnihd_affine_act_fwd_f32_planes")
            else:
                fwd = lib().omnihd_affine_act_fwd_f32 if x.dtype == torch.float32 else lib().omnihd_affine_act_fwd
                check(fwd(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), None if res is None else res.data_ptr(), y.data_ptr(),
                          n * h * w, c, 1 if relu else 0, _raw_stream()), "omnihd_affine_act_fwd")
        ctx.save_for_backward(y if relu else None, scale)
        ctx.relu, ctx.has_res, ctx.dtype, ctx.f16 = relu, res is not None, x.dtype, f16
        if y16 is not None:
            tag_half(y, y16, pkey)
        elif y_planes is not None:
            tag_planes(y, y_planes, pkey)
        elif pkey is not None:
            tag_producer(y, pkey)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, scale = ctx.saved_tensors
        gy = gy.to(ctx.dtype).contiguous(memory_format=torch.channels_last)
        n, c, h, w = gy.shape
        gx = torch.empty_like(gy)
        gres = torch.empty_like(gy) if ctx.has_res and ctx.needs_input_grad[3] else None
        with _on(gy.device):
            if ctx.f16 and gy.dtype == torch.float32 and f16_handover():
                # the convolution in front of this layer casts gx to half with a scale: max |gx| is accumulated here, on the way
                slot = _amax_slot(gy.device)
                check(lib().omnihd_affine_act_bwd_f32_amax(gy.data_ptr(), None if y is None else y.data_ptr(), scale.data_ptr(), gx.data_ptr(),
                                                           None if gres is None else gres.data_ptr(), slot.data_ptr(), n * h * w, c,
                                                           1 if ctx.relu else 0, _raw_stream()), "omnihd_affine_act_bwd_f32_amax")
                gx._omnihd_amax = (slot, gx._version)
            else:
                bwd = lib().omnihd_affine_act_bwd_f32 if gy.dtype == torch.float32 else lib().omnihd_affine_act_bwd
                check(bwd(gy.data_ptr(), None if y is None else y.data_ptr(), scale.data_ptr(), gx.data_ptr(),
                          None if gres is None else gres.data_ptr(), n * h * w, c, 1 if ctx.relu else 0, _raw_stream()),
                      "omnihd_affine_act_bwd")
        return gx, None, None, gres, None


def affine_act_supported(x, res=None):
    ok = x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32) and x.shape[1] % 8 == 0
    return ok and (res is None or (res.shape == x.shape and res.dtype == x.dtype and res.is_cuda))


def affine_act(x, scale, shift, res=None, relu=True):
    """x, res (N,C,H,W) bf16 or fp32 (made channels-last if they are not), scale/shift (C,) fp32 constants."""
    x = x.contiguous(memory_format=torch.channels_last)
    if res is not None:
        res = res.contiguous(memory_format=torch.channels_last)
    _want(scale, torch.float32, "scale")
    _want(shift, torch.float32, "shift")
    return _AffineAct.apply(x, scale, shift, res, relu)


class _BnTrainAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, group, res, unbiased_sync=False,
                grad_planes_only=False):
        import torch.distributed as dist
        ctx.grad_planes_only = bool(grad_planes_only)
        rows, c = _rows_view(x)
        dev = x.device
        ranks = dist.get_world_size(group) if group is not None else 1
        sfx = "_f32" if x.dtype == torch.float32 else ""
        buf = torch.empty(6, c, dtype=torch.float32, device=dev)           # [0:2] statistics, [2:6] scale, shift, mean, invstd
        stats, consts = buf[:2].view(-1), buf[2:]
        y = torch.empty_like(x)
        gamma, beta = _f32c(weight), _f32c(bias)
        L = lib()
        st = _raw_stream()
        # fp32 4-D outputs can be handed to the next split convolution as planes (see take_planes)
        pkey = ("bn_y", id(weight)) if (x.dtype == torch.float32 and x.dim() == 4 and ranks == 1) else None
        y_planes = _alloc_planes(y) if (pkey is not None and planes_wanted(pkey)) else None
        if y_planes is None and pkey is not None and half_wanted(pkey):
            y_planes = (torch.empty_like(y, dtype=torch.float16),)      # TF32-grade neighbour: the half plane of y (see take_half)
        rm = None if running_mean is None else running_mean.data_ptr()
        rv = None if running_var is None else running_var.data_ptr()
        resp = None if res is None else res.data_ptr()
        with _on(dev):
            nbytes = _SIZE_CACHE.get(("bn", rows, c))
            if nbytes is None:
                nbytes = _SIZE_CACHE[("bn", rows, c)] = L.omnihd_bn_workspace_bytes(rows, c)
            ws = _wgrad_workspace(nbytes, dev)
            if ranks == 1:
                # torch's BatchNorm keeps the unbiased variance in running_var
                corr = rows / (rows - 1.0) if rows > 1 else 1.0
                if y_planes is not None:
                    check(L.omnihd_bn_train_fwd_f32_planes(
                        x.data_ptr(), resp, gamma.data_ptr(), beta.data_ptr(), rm, rv, momentum, eps, corr, 1 if relu else 0,
                        y.data_ptr(), y_planes[0].data_ptr(), y_planes[1].data_ptr() if len(y_planes) == 2 else None, stats.data_ptr(),
                        consts.data_ptr(), rows, c, ws.data_ptr(), ws.numel(), st), "omnihd_bn_train_fwd_f32_planes")
                else:
                    check(getattr(L, "omnihd_bn_train_fwd" + sfx)(
                        x.data_ptr(), resp, gamma.data_ptr(), beta.data_ptr(), rm, rv, momentum, eps, corr, 1 if relu else 0,
                        y.data_ptr(), stats.data_ptr(), consts.data_ptr(), rows, c, ws.data_ptr(), ws.numel(), st), "omnihd_bn_train_fwd")
            else:
                check(getattr(L, "omnihd_bn_channel_sums" + sfx)(x.data_ptr(), None, None, None, stats.data_ptr(), rows, c, 0,
                                                                 1.0 / rows, ws.data_ptr(), ws.numel(), st), "omnihd_bn_channel_sums")
                dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
                # the reference's SyncBN keeps the biased variance (ops/norm.py:74-75); torch's SyncBatchNorm the unbiased
                # one over the rows of all ranks
                total = float(ranks * rows)
                corr = total / (total - 1.0) if (unbiased_sync and total > 1) else 1.0
                check(L.omnihd_bn_fwd_consts(stats.data_ptr(), 1.0 / ranks, gamma.data_ptr(), beta.data_ptr(), eps, momentum, corr,
                                             c, rm, rv, consts[0].data_ptr(), consts[1].data_ptr(), consts[2].data_ptr(),
                                             consts[3].data_ptr(), st), "omnihd_bn_fwd_consts")
                check(getattr(L, "omnihd_affine_act_fwd" + sfx)(x.data_ptr(), consts[0].data_ptr(), consts[1].data_ptr(), resp,
                                                                y.data_ptr(), rows, c, 1 if relu else 0, st), "omnihd_affine_act_fwd")
        # The ReLU mask of the backward comes from the saved output.  The kernels can also recompute it from x with the
        # forward's constants (OMNIHD_BN_MASK_FROM_X=1: one tensor less to read), but that measured SLOWER in the full
        # step (34.1-35.0 vs 32.5-33.3 ms, alternating blocks in one process): the per-element constant loads cost more
        # than the streamed read they save.
        keep_y = relu and (res is not None or _env("OMNIHD_BN_MASK_FROM_X", "0") != "1")
        ctx.save_for_backward(x, y if keep_y else None, gamma, consts)
        ctx.relu, ctx.group, ctx.ranks, ctx.param_dtypes = relu, group, ranks, (weight.dtype, bias.dtype)
        ctx.has_res = res is not None
        ctx.gkey = ("bn_gx", id(weight)) if pkey is not None else None
        if pkey is not None:
            if y_planes is not None and len(y_planes) == 1:
                tag_half(y, y_planes[0], pkey)
            elif y_planes is not None:
                tag_planes(y, y_planes, pkey)
            else:
                tag_producer(y, pkey)
        return y

    @staticmethod
    def backward(ctx, gy):
        import torch.distributed as dist
        x, y, gamma, consts = ctx.saved_tensors
        gy = gy.to(x.dtype)
        gy = gy.contiguous(memory_format=torch.channels_last) if gy.dim() == 4 else gy.contiguous()
        sfx = "_f32" if x.dtype == torch.float32 else ""
        rows, c = _rows_view(x)
        dev = x.device
        buf = torch.empty(7, c, dtype=torch.float32, device=dev)           # [0:2] sums, [2:7] dgamma, dbeta, A, B, C
        local, out = buf[:2].view(-1), buf[2:]
        gx = torch.empty_like(x)
        gres = None
        if ctx.has_res and ctx.needs_input_grad[9]:
            gres = torch.empty_like(x) if ctx.relu else gy          # without a ReLU the residual's gradient is gy itself
        gresp = gres.data_ptr() if (gres is not None and ctx.relu) else None
        yp = None if y is None else y.data_ptr()
        L = lib()
        st = _raw_stream()
        with _on(dev):
            ws = _wgrad_workspace(_SIZE_CACHE[("bn", rows, c)], dev)
            if ctx.grad_planes_only and ctx.ranks == 1 and x.dtype == torch.float32 and x.numel() % 8 == 0:
                # The convolution in front of this layer is the ONLY consumer of gx (conv_bn_act keeps the tensor between them
                # to itself) and reads it as hi / lo planes: write the planes only.  The fp32-typed tensor autograd carries
                # between the two nodes is a view of the plane buffer (same byte count) — its fp32 values are never read.
                n = x.numel()
                buf = torch.empty(2 * n, dtype=torch.bfloat16, device=dev)
                gx_planes = tuple(buf[i * n:(i + 1) * n].as_strided(x.shape, x.stride()) for i in (0, 1))
                gx = buf.view(torch.float32).as_strided(x.shape, x.stride())
                check(L.omnihd_bn_train_bwd_f32_planes(
                    gy.data_ptr(), yp, 1 if ctx.relu else 0, x.data_ptr(), gamma.data_ptr(), consts.data_ptr(), None,
                    gx_planes[0].data_ptr(), gx_planes[1].data_ptr(), gresp, local.data_ptr(), out.data_ptr(), rows, c,
                    ws.data_ptr(), ws.numel(), st), "omnihd_bn_train_bwd_f32_planes")
                gx._omnihd_planes = (gx_planes, gx._version, ("bn_gx_only", 0))
                gx._omnihd_planes_only = True
                FAST_PATHS["grad_planes_only"] = FAST_PATHS.get("grad_planes_only", 0) + 1
                return (gx, out[0].to(ctx.param_dtypes[0]), out[1].to(ctx.param_dtypes[1]), None, None, None, None, None, None,
                        gres, None, None)
            gx_planes = _alloc_planes(gx) if (ctx.gkey is not None and ctx.ranks == 1 and planes_wanted(ctx.gkey)) else None
            if gx_planes is not None:
                check(L.omnihd_bn_train_bwd_f32_planes(
                    gy.data_ptr(), yp, 1 if ctx.relu else 0, x.data_ptr(), gamma.data_ptr(), consts.data_ptr(), gx.data_ptr(),
                    gx_planes[0].data_ptr(), gx_planes[1].data_ptr(), gresp, local.data_ptr(), out.data_ptr(), rows, c,
                    ws.data_ptr(), ws.numel(), st), "omnihd_bn_train_bwd_f32_planes")
                tag_planes(gx, gx_planes, ctx.gkey)
            elif ctx.ranks == 1 and ctx.gkey is not None and f16_handover():
                # the TF32-grade convolution in front of this layer casts gx to half with a scale: max |gx| accumulated on the way
                slot = _amax_slot(dev)
                check(L.omnihd_bn_train_bwd_f32_amax(
                    gy.data_ptr(), yp, 1 if ctx.relu else 0, x.data_ptr(), gamma.data_ptr(), consts.data_ptr(), gx.data_ptr(),
                    slot.data_ptr(), gresp, local.data_ptr(), out.data_ptr(), rows, c, ws.data_ptr(), ws.numel(), st),
                    "omnihd_bn_train_bwd_f32_amax")
                gx._omnihd_amax = (slot, gx._version)
            elif ctx.ranks == 1:
                if ctx.gkey is not None:
                    tag_producer(gx, ctx.gkey)
                check(getattr(L, "omnihd_bn_train_bwd" + sfx)(
                    gy.data_ptr(), yp, 1 if ctx.relu else 0, x.data_ptr(), gamma.data_ptr(), consts.data_ptr(), gx.data_ptr(),
                    gresp, local.data_ptr(), out.data_ptr(), rows, c, ws.data_ptr(), ws.numel(), st), "omnihd_bn_train_bwd")
            else:
                fss = consts.data_ptr() if (ctx.relu and yp is None) else None
                check(getattr(L, "omnihd_bn_channel_sums" + sfx)(gy.data_ptr(), x.data_ptr(), yp, fss, local.data_ptr(), rows, c,
                                                                 1, 1.0, ws.data_ptr(), ws.numel(), st), "omnihd_bn_channel_sums")
                glob = local.clone()
                dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=ctx.group)
                check(L.omnihd_bn_bwd_consts(local.data_ptr(), glob.data_ptr(), gamma.data_ptr(), consts[2].data_ptr(),
                                             consts[3].data_ptr(), 1.0 / (ctx.ranks * rows), c, out[0].data_ptr(),
                                             out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), out[4].data_ptr(), st),
                      "omnihd_bn_bwd_consts")
                check(getattr(L, "omnihd_bn_bwd_apply" + sfx)(gy.data_ptr(), yp, fss, x.data_ptr(), out[2].data_ptr(),
                                                              out[3].data_ptr(), out[4].data_ptr(), gx.data_ptr(), gresp, rows, c,
                                                              st), "omnihd_bn_bwd_apply")
        return (gx, out[0].to(ctx.param_dtypes[0]), out[1].to(ctx.param_dtypes[1]), None, None, None, None, None, None,
                gres, None, None)


def bn_train_supported(x):
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() in (2, 4) and x.shape[1] % 8 == 0
            and x.shape[1] <= 2048):
        return False
    return x.numel() > 0


def bn_train_act(x, weight, bias, running_mean, running_var, momentum, eps, relu=False, group=None, residual=None,
                 unbiased_sync=False, grad_planes_only=False):
    """``act(BatchNorm_train(x) + residual)`` of a bf16 or fp32 (N,C,H,W) [made channels-last] or (N,C) tensor; statistics
    are the mean over ``group``'s ranks of the per-rank mean / mean of squares when a group with more than one rank is
    given (``unbiased_sync``: running_var takes the unbiased variance over all ranks' rows, as torch's SyncBatchNorm)."""
    cl = (lambda t: t.contiguous(memory_format=torch.channels_last)) if x.dim() == 4 else (lambda t: t.contiguous())
    return _BnTrainAct.apply(cl(x), weight, bias, running_mean, running_var, float(momentum), float(eps), bool(relu), group,
                             None if residual is None else cl(residual), bool(unbiased_sync), bool(grad_planes_only))
