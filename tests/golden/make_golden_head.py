#!/usr/bin/env python3
"""Golden for the per-level loss of the anchor head the reference VENDORS
(projects/mmdet3d_plugin/bevfusion/dense_heads/det_anchor3d_head.py:192-301, ``Anchor3DHeadV1.loss_single`` and
``add_sin_difference``), run unbound on a shell ``self`` over synthetic maps and targets.

The three loss callables come from upstream mmdet (absent); their stand-ins here restate the published formulas and
carry arithmetic: sigmoid focal loss (gamma, alpha, one-hot targets with the background index = num_classes giving an
all-zero row), smooth-L1 with ``beta``, softmax cross-entropy; each ``sum(loss * weight) / avg_factor * loss_weight``.
What the fixture pins is the head's own code: the permutes/reshapes of the three maps, the positive selection, code
weights, the sine encoding of the yaw residual and the use of ``num_total_samples``.
Usage: python tests/golden/make_golden_head.py"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_data as G  # noqa: E402


def focal(pred, target, weight=None, avg_factor=None, gamma=2.0, alpha=0.25, loss_weight=1.0):
    t = F.one_hot(target, num_classes=pred.shape[1] + 1)[:, :pred.shape[1]].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    fw = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, t, reduction="none") * fw
    return loss_weight * (loss * weight.view(-1, 1)).sum() / avg_factor


def smooth_l1(pred, target, weight=None, avg_factor=None, beta=1.0 / 9.0, loss_weight=1.0):
    d = (pred - target).abs()
    loss = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
    return loss_weight * (loss * weight).sum() / avg_factor


def cross_entropy(pred, label, weight=None, avg_factor=None, loss_weight=0.2):
    return loss_weight * (F.cross_entropy(pred, label, reduction="none") * weight).sum() / avg_factor


def main():
    G.install_stubs()
    passthrough = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))      # noqa: E731
    G._mod("mmcv.runner", BaseModule=torch.nn.Module, force_fp32=passthrough, auto_fp16=passthrough)
    core3d = sys.modules["mmdet3d.core"]
    for n in ("PseudoSampler", "box3d_multiclass_nms", "limit_period", "xywhr2xyxyr"):
        setattr(core3d, n, None)
    G._mod("mmdet.core", build_assigner=None, build_bbox_coder=None, build_anchor_generator=None, build_sampler=None,
           build_prior_generator=None, multi_apply=None)
    G._mod("mmdet.models", HEADS=G._Registry())
    G._mod("mmdet3d.models")
    G._mod("mmdet3d.models.builder", HEADS=G._Registry(), build_loss=None)
    G._mod("mmdet3d.models.dense_heads")
    G._mod("mmdet3d.models.dense_heads.train_mixins", AnchorTrainMixin=object)
    sys.path.insert(0, G.REF)
    mod = G.load_file("ref_anchor_head", "projects/mmdet3d_plugin/bevfusion/dense_heads/det_anchor3d_head.py")
    Head = mod.Anchor3DHeadV1
    rng = np.random.default_rng(41)
    out = {}
    B, A, K, H, W, C = 2, 4, 3, 5, 6, 9                       # anchors per cell, classes, map size, box code size
    n = B * H * W * A
    cls = torch.from_numpy(rng.normal(size=(B, A * K, H, W)).astype(np.float32))
    box = torch.from_numpy(rng.normal(size=(B, A * C, H, W)).astype(np.float32))
    dirs = torch.from_numpy(rng.normal(size=(B, A * 2, H, W)).astype(np.float32))
    labels = torch.from_numpy(np.where(rng.random(n) < 0.15, rng.integers(0, K, n), K))      # K = background
    label_w = torch.from_numpy((rng.random(n) < 0.9).astype(np.float32))
    pos = labels < K
    box_t = torch.from_numpy(rng.normal(size=(n, C)).astype(np.float32)) * pos[:, None]
    box_w = pos[:, None].float().expand(n, C).contiguous()
    dir_t = torch.from_numpy(rng.integers(0, 2, n)) * pos
    dir_w = pos.float()
    for tag, cw, sin in (("cw_sin", [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2], True), ("plain", None, False)):
        shell = types.SimpleNamespace(num_classes=K, box_code_size=C, use_direction_classifier=True, diff_rad_by_sin=sin,
                                      train_cfg=dict(code_weight=cw), loss_cls=focal, loss_bbox=smooth_l1,
                                      loss_dir=cross_entropy, add_sin_difference=Head.add_sin_difference)
        num_total = int(pos.sum())
        lc, lb, ld = Head.loss_single(shell, cls, box, dirs, labels.view(B, -1), label_w.view(B, -1), box_t.view(B, -1, C),
                                      box_w.view(B, -1, C), dir_t.view(B, -1), dir_w.view(B, -1), num_total)
        out[f"{tag}_losses"] = np.array([float(lc), float(lb), float(ld)])
        print(tag, out[f"{tag}_losses"])
    for k, v in dict(cls=cls, box=box, dirs=dirs, labels=labels, label_w=label_w, box_t=box_t, box_w=box_w, dir_t=dir_t,
                     dir_w=dir_w).items():
        out[k] = v.numpy()
    out["num_total"] = np.array(int(pos.sum()))
    b1 = torch.from_numpy(rng.normal(size=(7, 9)).astype(np.float32))
    b2 = torch.from_numpy(rng.normal(size=(7, 9)).astype(np.float32))
    s1, s2 = Head.add_sin_difference(b1, b2)
    out["sin_b1"], out["sin_b2"], out["sin_o1"], out["sin_o2"] = b1.numpy(), b2.numpy(), s1.numpy(), s2.numpy()
    # ---- test-time branch: get_bboxes_single (:425-514) on a shell, two levels so that nms_pre cuts only one --------
    RECORD = {}

    class Boxes:                                       # what the head reads from ``input_meta['box_type_3d']``
        def __init__(self, tensor, box_dim=7):
            self.tensor = tensor
        bev = property(lambda self: self.tensor[:, [0, 1, 3, 4, 6]])

    def xywhr2xyxyr(b):                                # published formula of the absent helper
        o = torch.zeros_like(b)
        o[:, 0], o[:, 1], o[:, 2], o[:, 3], o[:, 4] = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, \
            b[:, 1] + b[:, 3] / 2, b[:, 4]
        return o

    def limit_period(val, offset=0.5, period=np.pi):
        return val - torch.floor(val / period + offset) * period

    def nms_recorder(bboxes, for_nms, scores, score_thr, max_num, cfg, dir_scores):
        """Stands in for the upstream NMS: records its inputs, keeps every 3rd candidate with its best class."""
        RECORD.update(bboxes=bboxes.clone(), for_nms=for_nms.clone(), scores=scores.clone(), score_thr=score_thr,
                      max_num=max_num, dir_scores=dir_scores.clone())
        keep = torch.arange(0, bboxes.shape[0], 3)[:max_num]
        best, lab = scores[keep, :-1].max(dim=1)
        return bboxes[keep], best, lab, dir_scores[keep]

    class Decode:                                      # DeltaXYZWLHRBBoxCoder.decode, published formula
        @staticmethod
        def decode(anchors, deltas):
            xa, ya, za, wa, la, ha, ra = torch.split(anchors[..., :7], 1, dim=-1)
            xt, yt, zt, wt, lt, ht, rt = torch.split(deltas[..., :7], 1, dim=-1)
            za = za + ha / 2
            diag = torch.sqrt(la ** 2 + wa ** 2)
            xg, yg, zg = xt * diag + xa, yt * diag + ya, zt * ha + za
            lg, wg, hg = torch.exp(lt) * la, torch.exp(wt) * wa, torch.exp(ht) * ha
            return torch.cat([xg, yg, zg - hg / 2, wg, lg, hg, rt + ra, deltas[..., 7:] + anchors[..., 7:]], dim=-1)

    mod.xywhr2xyxyr, mod.limit_period, mod.box3d_multiclass_nms = xywhr2xyxyr, limit_period, nms_recorder

    class Cfg(dict):
        __getattr__ = dict.__getitem__
    cfg = Cfg(nms_pre=40, score_thr=0.05, max_num=20, use_rotate_nms=True, nms_thr=0.2)
    shell = types.SimpleNamespace(test_cfg=cfg, num_classes=K, box_code_size=C, use_sigmoid_cls=True, bbox_coder=Decode,
                                  dir_offset=0.7854, dir_limit_offset=0)
    levels = [(5, 6), (2, 3)]
    cls_l = [torch.from_numpy(rng.normal(size=(A * K, h, w)).astype(np.float32)) for h, w in levels]
    box_l = [torch.from_numpy((0.3 * rng.normal(size=(A * C, h, w))).astype(np.float32)) for h, w in levels]
    dir_l = [torch.from_numpy(rng.normal(size=(A * 2, h, w)).astype(np.float32)) for h, w in levels]
    anc_l = []
    for h, w in levels:
        a = rng.normal(size=(h * w * A, C)).astype(np.float32)
        a[:, 3:6] = np.abs(a[:, 3:6]) + 0.5
        anc_l.append(torch.from_numpy(a))
    boxes, scores, labels = Head.get_bboxes_single(shell, cls_l, box_l, dir_l, anc_l, dict(box_type_3d=Boxes))
    for i in range(2):
        out[f"tt_cls{i}"], out[f"tt_box{i}"], out[f"tt_dir{i}"], out[f"tt_anc{i}"] = cls_l[i].numpy(), box_l[i].numpy(), \
            dir_l[i].numpy(), anc_l[i].numpy()
    out["tt_boxes"], out["tt_scores"], out["tt_labels"] = boxes.tensor.numpy(), scores.numpy(), labels.numpy()
    for k in ("bboxes", "for_nms", "scores", "dir_scores"):
        out["tt_nms_in_" + k] = RECORD[k].numpy()
    out["tt_nms_in_thr_max"] = np.array([RECORD["score_thr"], RECORD["max_num"]])
    print("get_bboxes_single:", tuple(boxes.tensor.shape), "candidates into the NMS:", RECORD["bboxes"].shape[0])

    path = os.path.join(HERE, "head_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
