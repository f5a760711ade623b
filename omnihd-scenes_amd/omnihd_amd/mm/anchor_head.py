"""Anchor3DHead and its helpers (anchor generator, box coder, assigner, losses) with the
mmdet3d v0.17.1 / mmdet 2.14 names the reference config uses
(projects/configs/bevfusion_NewScenes/bevfusion.py:96-155).

Upstream is not vendored in the reference; the loss/decoding logic is readable there only through
the vendored copy ``Anchor3DHeadV1`` (projects/mmdet3d_plugin/bevfusion/dense_heads/
det_anchor3d_head.py: loss_single :192-276, add_sin_difference :278-301, loss :303-372), which this
file follows; anchor generation, target assignment and the box coder are restated from the
upstream release's documented behaviour (parity unpinned by the reference; pinned by
tests/test_model_cpu.py known-answer cases).  Boxes are (x, y, z_bottom, w, l, h, yaw, vx, vy) in
the LiDAR frame.  Everything is dense torch tensor math on the device (no host loops over anchors).
"""
from .._env import env as _env
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .registry import HEADS, LOSSES, Registry, build_from_cfg

ANCHOR_GENERATORS = Registry("anchor generator")
BBOX_CODERS = Registry("bbox coder")
BBOX_ASSIGNERS = Registry("bbox assigner")
IOU_CALCULATORS = Registry("iou calculator")


def limit_period(val, offset=0.5, period=math.pi):
    return val - torch.floor(val / period + offset) * period


def _gt_tensor(b):
    return b.tensor if hasattr(b, "tensor") else b


# ---------------------------------------------------------------------------------------------
@ANCHOR_GENERATORS.register_module()
class AlignedAnchor3DRangeGenerator:
    """Anchors at the CENTRES of the feature-map cells; output order (z, y, x, size, rotation)."""

    def __init__(self, ranges, sizes=((1.6, 3.9, 1.56), ), scales=(1, ), rotations=(0, 1.5707963), custom_values=(),
                 reshape_out=True, size_per_range=True, align_corner=False):
        if len(ranges) != len(sizes):
            assert len(ranges) == 1
            ranges = list(ranges) * len(sizes)
        self.ranges, self.sizes, self.scales = ranges, sizes, scales
        self.rotations, self.custom_values = rotations, custom_values
        self.reshape_out, self.align_corner = reshape_out, align_corner
        self._cache = {}

    @property
    def num_base_anchors(self):
        return len(self.rotations) * len(self.sizes)

    @property
    def num_levels(self):
        return len(self.scales)

    def anchors_single_range(self, feature_size, anchor_range, scale, sizes, rotations, device):
        if len(feature_size) == 2:
            feature_size = [1, feature_size[0], feature_size[1]]
        r = torch.tensor(anchor_range, device=device, dtype=torch.float32)
        zc = torch.linspace(r[2], r[5], feature_size[0] + 1, device=device)
        yc = torch.linspace(r[1], r[4], feature_size[1] + 1, device=device)
        xc = torch.linspace(r[0], r[3], feature_size[2] + 1, device=device)
        sizes = torch.tensor(sizes, device=device, dtype=torch.float32).reshape(-1, 3) * scale
        rotations = torch.tensor(rotations, device=device, dtype=torch.float32)
        if not self.align_corner:
            zc = zc + (zc[1] - zc[0]) / 2
            yc = yc + (yc[1] - yc[0]) / 2
            xc = xc + (xc[1] - xc[0]) / 2
        X, Y, Z, R = torch.meshgrid(xc[:feature_size[2]], yc[:feature_size[1]], zc[:feature_size[0]], rotations,
                                    indexing="ij")
        ns = sizes.shape[0]
        parts = [t.unsqueeze(-2).repeat(1, 1, 1, ns, 1).unsqueeze(-1) for t in (X, Y, Z, R)]
        S = sizes.reshape(1, 1, 1, ns, 1, 3).repeat(X.shape[0], X.shape[1], X.shape[2], 1, X.shape[3], 1)
        ret = torch.cat([parts[0], parts[1], parts[2], S, parts[3]], dim=-1).permute(2, 1, 0, 3, 4, 5)
        if len(self.custom_values) > 0:
            ret = torch.cat([ret, ret.new_zeros([*ret.shape[:-1], len(self.custom_values)])], dim=-1)
        return ret                                        # [D, H, W, sizes, rots, 7 + custom]

    def grid_anchors(self, featmap_sizes, device="cuda"):
        out = []
        for i, fs in enumerate(featmap_sizes):
            key = (tuple(fs), str(device), i)
            if key not in self._cache:
                per_range = [self.anchors_single_range(fs, rng, self.scales[i], size, self.rotations, device)
                             for rng, size in zip(self.ranges, self.sizes)]
                a = torch.cat(per_range, dim=-3)
                if self.reshape_out:
                    a = a.reshape(-1, a.size(-1))
                self._cache[key] = a
            out.append(self._cache[key])
        return out


@BBOX_CODERS.register_module()
class DeltaXYZWLHRBBoxCoder:
    def __init__(self, code_size=7):
        self.code_size = code_size

    @staticmethod
    def encode(src, dst):
        xa, ya, za, wa, la, ha, ra = [src[..., i] for i in range(7)]
        xg, yg, zg, wg, lg, hg, rg = [dst[..., i] for i in range(7)]
        za = za + ha / 2
        zg = zg + hg / 2
        diag = torch.sqrt(la ** 2 + wa ** 2)
        code = [(xg - xa) / diag, (yg - ya) / diag, (zg - za) / ha, torch.log(wg / wa), torch.log(lg / la),
                torch.log(hg / ha), rg - ra]
        code += [dst[..., i] - src[..., i] for i in range(7, src.shape[-1])]
        return torch.stack(code, dim=-1)

    @staticmethod
    def decode(anchors, deltas):
        xa, ya, za, wa, la, ha, ra = [anchors[..., i] for i in range(7)]
        xt, yt, zt, wt, lt, ht, rt = [deltas[..., i] for i in range(7)]
        za = za + ha / 2
        diag = torch.sqrt(la ** 2 + wa ** 2)
        hg = torch.exp(ht) * ha
        out = [xt * diag + xa, yt * diag + ya, zt * ha + za - hg / 2, torch.exp(wt) * wa, torch.exp(lt) * la, hg,
               rt + ra]
        out += [deltas[..., i] + anchors[..., i] for i in range(7, anchors.shape[-1])]
        return torch.stack(out, dim=-1)


@IOU_CALCULATORS.register_module()
class BboxOverlapsNearest3D:
    """IoU of the nearest axis-aligned BEV boxes (rotation snapped to 0 / 90 degrees)."""

    def __init__(self, coordinate="lidar"):
        self.coordinate = coordinate

    @staticmethod
    def nearest_bev(b):
        rot = torch.abs(limit_period(b[:, 6], 0.5, math.pi))
        swap = (rot > math.pi / 4)[:, None]
        dims = torch.where(swap, b[:, 3:5].flip(-1), b[:, 3:5])          # no index lists: they cost a host->device copy
        c = b[:, :2]
        return torch.cat([c - dims / 2, c + dims / 2], dim=-1)

    def __call__(self, b1, b2, mode="iou", is_aligned=False):
        a, b = self.nearest_bev(b1), self.nearest_bev(b2)
        area1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
        area2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        lt = torch.max(a[:, None, :2], b[None, :, :2])
        rb = torch.min(a[:, None, 2:], b[None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = (area1[:, None] + area2[None, :] - overlap).clamp(min=1e-6)
        return overlap / union


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1,
                 iou_calculator=dict(type="BboxOverlaps2D")):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all, self.match_low_quality = gt_max_assign_all, match_low_quality
        self.iou_calculator = build_from_cfg(iou_calculator, IOU_CALCULATORS)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        """-> assigned_gt_inds (N,) long: -1 ignore, 0 negative, k>0 matched to gt k-1."""
        n = bboxes.shape[0]
        assigned = bboxes.new_full((n,), -1, dtype=torch.long)
        if gt_bboxes.shape[0] == 0:
            assigned[:] = 0
            return assigned
        overlaps = self.iou_calculator(gt_bboxes, bboxes)            # (num_gt, N)
        max_ov, argmax_ov = overlaps.max(dim=0)
        gt_max_ov, gt_argmax = overlaps.max(dim=1)
        # masks + torch.where instead of boolean-mask indexing: no host synchronisation anywhere in the assigner
        assigned = torch.where((max_ov >= 0) & (max_ov < self.neg_iou_thr), torch.zeros_like(assigned), assigned)
        assigned = torch.where(max_ov >= self.pos_iou_thr, argmax_ov + 1, assigned)
        if self.match_low_quality:
            # for gt i (in order; later gts overwrite earlier ones): every anchor reaching gt i's best IoU
            ok = gt_max_ov >= self.min_pos_iou
            if self.gt_max_assign_all:
                hit = (overlaps == gt_max_ov[:, None]) & ok[:, None]                  # (num_gt, N)
                idx = torch.arange(1, gt_bboxes.shape[0] + 1, device=bboxes.device)[:, None].expand_as(hit)
                last = torch.where(hit, idx, torch.zeros_like(idx)).max(dim=0)[0]
                assigned = torch.where(last > 0, last, assigned)
            else:
                assigned[gt_argmax[ok]] = torch.nonzero(ok).flatten() + 1
        return assigned


# ---------------------------------------------------------------------------------------------
def _reduce(loss, weight, avg_factor, reduction="mean"):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == "mean" else loss.sum()
    return loss.sum() / avg_factor


@LOSSES.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0):
        super().__init__()
        assert use_sigmoid
        self.gamma, self.alpha, self.loss_weight = gamma, alpha, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        nc = pred.size(1)
        t = F.one_hot(target.clamp(max=nc), nc + 1)[:, :nc].type_as(pred)     # label == nc is background
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        fw = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction="none") * fw
        if weight is not None:
            weight = weight.view(-1, 1)
        return self.loss_weight * _reduce(loss, weight, avg_factor)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.beta, self.loss_weight = beta, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None):
        diff = torch.abs(pred - target)
        loss = torch.where(diff < self.beta, 0.5 * diff * diff / self.beta, diff - 0.5 * self.beta)
        return self.loss_weight * _reduce(loss, weight, avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, reduction="mean", loss_weight=1.0, **_):
        super().__init__()
        assert not use_sigmoid
        self.loss_weight = loss_weight

    def forward(self, pred, label, weight=None, avg_factor=None):
        loss = F.cross_entropy(pred, label, reduction="none")
        return self.loss_weight * _reduce(loss, weight, avg_factor)


# ---------------------------------------------------------------------------------------------
@HEADS.register_module()
class Anchor3DHead(nn.Module):
    def __init__(self, num_classes, in_channels, train_cfg=None, test_cfg=None, feat_channels=256,
                 use_direction_classifier=True, anchor_generator=None, assigner_per_size=False,
                 assign_per_class=False, diff_rad_by_sin=True, dir_offset=0, dir_limit_offset=1,
                 bbox_coder=dict(type="DeltaXYZWLHRBBoxCoder"), loss_cls=None, loss_bbox=None, loss_dir=None, **_):
        super().__init__()
        self.in_channels, self.num_classes, self.feat_channels = in_channels, num_classes, feat_channels
        self.diff_rad_by_sin, self.use_direction_classifier = diff_rad_by_sin, use_direction_classifier
        self.train_cfg, self.test_cfg = train_cfg or {}, test_cfg
        self.assigner_per_size, self.assign_per_class = assigner_per_size, assign_per_class
        self.dir_offset, self.dir_limit_offset = dir_offset, dir_limit_offset
        self.anchor_generator = build_from_cfg(anchor_generator, ANCHOR_GENERATORS)
        self.num_anchors = self.anchor_generator.num_base_anchors
        self.bbox_coder = build_from_cfg(bbox_coder, BBOX_CODERS)
        self.box_code_size = self.bbox_coder.code_size
        self.use_sigmoid_cls = loss_cls.get("use_sigmoid", False)
        self.sampling = loss_cls["type"] not in ["FocalLoss", "GHMC"]
        self.loss_cls = build_from_cfg(loss_cls, LOSSES)
        self.loss_bbox = build_from_cfg(loss_bbox, LOSSES)
        self.loss_dir = build_from_cfg(loss_dir, LOSSES)
        assert not self.sampling and not assigner_per_size, "only the configuration of the reference is restated"
        if self.train_cfg:
            self.bbox_assigner = build_from_cfg(self.train_cfg["assigner"], BBOX_ASSIGNERS)
        self.cls_out_channels = self.num_anchors * num_classes
        self.conv_cls = nn.Conv2d(feat_channels, self.cls_out_channels, 1)
        self.conv_reg = nn.Conv2d(feat_channels, self.num_anchors * self.box_code_size, 1)
        if use_direction_classifier:
            self.conv_dir_cls = nn.Conv2d(feat_channels, self.num_anchors * 2, 1)
        for m in (self.conv_cls, self.conv_reg):
            nn.init.normal_(m.weight, std=0.01)
        nn.init.constant_(self.conv_cls.bias, float(-np.log((1 - 0.01) / 0.01)))
        nn.init.constant_(self.conv_reg.bias, 0)

    def forward(self, feats):
        cls, reg, dirs = [], [], []
        for x in feats:
            cls.append(self.conv_cls(x))
            reg.append(self.conv_reg(x))
            dirs.append(self.conv_dir_cls(x) if self.use_direction_classifier else None)
        return cls, reg, dirs

    # ---- targets ----------------------------------------------------------------------------
    def _targets_single(self, anchors, gt_bboxes, gt_labels):
        """Dense, mask-based target assignment: every quantity is computed for all anchors and selected with
        ``torch.where`` — same values as gathering the positives, but no ``nonzero`` and hence no host sync."""
        n = anchors.shape[0]
        gt = _gt_tensor(gt_bboxes).to(anchors.device).float()
        if gt.shape[0] == 0:
            zeros = anchors.new_zeros(n)
            return (anchors.new_full((n,), self.num_classes, dtype=torch.long), anchors.new_ones(n), torch.zeros_like(anchors),
                    torch.zeros_like(anchors), zeros.long(), zeros, zeros.sum())
        assigned = self.bbox_assigner.assign(anchors, gt, None, gt_labels)
        pos, neg = assigned > 0, assigned == 0
        gi = (assigned - 1).clamp(min=0)
        t = self.bbox_coder.encode(anchors, gt[gi])
        posc = pos.unsqueeze(-1)
        bbox_targets = torch.where(posc, t, torch.zeros_like(t))
        bbox_weights = posc.to(anchors.dtype).expand_as(anchors)
        rot_gt = t[..., 6] + anchors[..., 6]
        offset_rot = limit_period(rot_gt - self.dir_offset, 0, 2 * math.pi)
        dir_targets = torch.where(pos, torch.floor(offset_rot / math.pi).long().clamp(0, 1), torch.zeros_like(assigned))
        dir_weights = pos.to(anchors.dtype)
        labels = torch.where(pos, gt_labels.to(anchors.device).long()[gi], torch.full_like(assigned, self.num_classes))
        label_weights = (pos | neg).to(anchors.dtype)
        return labels, label_weights, bbox_targets, bbox_weights, dir_targets, dir_weights, pos.sum()

    @staticmethod
    def add_sin_difference(b1, b2):
        r1 = torch.sin(b1[..., 6:7]) * torch.cos(b2[..., 6:7])
        r2 = torch.cos(b1[..., 6:7]) * torch.sin(b2[..., 6:7])
        return (torch.cat([b1[..., :6], r1, b1[..., 7:]], dim=-1), torch.cat([b2[..., :6], r2, b2[..., 7:]], dim=-1))

    def loss(self, cls_scores, bbox_preds, dir_cls_preds, gt_bboxes, gt_labels, input_metas, gt_bboxes_ignore=None):
        """Same arithmetic as the vendored head's ``loss`` / ``loss_single`` (det_anchor3d_head.py:192-372);
        one feature level (the config has a single scale)."""
        assert len(cls_scores) == 1
        fused = self._fused_loss(cls_scores[0], bbox_preds[0], dir_cls_preds[0], gt_bboxes, gt_labels)
        if fused is not None:
            return fused
        cls_score, bbox_pred, dir_pred = cls_scores[0].float(), bbox_preds[0].float(), dir_cls_preds[0].float()
        B = cls_score.shape[0]
        anchors = self.anchor_generator.grid_anchors([cls_score.shape[-2:]], device=cls_score.device)[0]
        tg = [self._targets_single(anchors, gt_bboxes[i], gt_labels[i]) for i in range(B)]
        labels = torch.stack([t[0] for t in tg]).reshape(-1)
        label_weights = torch.stack([t[1] for t in tg]).reshape(-1)
        bbox_targets = torch.stack([t[2] for t in tg]).reshape(-1, self.box_code_size)
        bbox_weights = torch.stack([t[3] for t in tg]).reshape(-1, self.box_code_size)
        dir_targets = torch.stack([t[4] for t in tg]).reshape(-1)
        dir_weights = torch.stack([t[5] for t in tg]).reshape(-1)
        num_total_samples = torch.stack([t[6] for t in tg]).clamp(min=1).sum().to(cls_score.dtype)   # stays on the device
        return self.loss_from_targets(cls_score, bbox_pred, dir_pred, labels, label_weights, bbox_targets, bbox_weights,
                                      dir_targets, dir_weights, num_total_samples)

    def _fused_loss(self, cls_score, bbox_pred, dir_pred, gt_bboxes, gt_labels):
        """The same losses from the fused kernels of csrc/anchor_loss.hip (target assignment + three losses + the gradient maps in
        three launches instead of ~150), where the head is configured as the reference configures it; None = not applicable (the
        torch formulation below runs).  OMNIHD_ANCHOR_LOSS=0 turns it off."""
        import os
        if _env("OMNIHD_ANCHOR_LOSS", "1") == "0" or not cls_score.is_cuda or not self.use_direction_classifier:
            return None
        asg = self.bbox_assigner
        if not (isinstance(self.loss_cls, FocalLoss) and isinstance(self.loss_bbox, SmoothL1Loss) and isinstance(self.loss_dir, CrossEntropyLoss)
                and isinstance(asg, MaxIoUAssigner) and isinstance(asg.iou_calculator, BboxOverlapsNearest3D)
                and asg.match_low_quality and asg.gt_max_assign_all and type(self.bbox_coder) is DeltaXYZWLHRBBoxCoder
                and self.num_classes <= 8 and 7 <= self.box_code_size <= 12):
            return None
        from .. import ops
        dev = cls_score.device
        gts = [_gt_tensor(g).to(dev).float().reshape(-1, self.box_code_size) for g in gt_bboxes]
        if any(g.shape[0] > 128 for g in gts):
            return None
        offs = np.zeros(len(gts) + 1, dtype=np.int32)
        offs[1:] = np.cumsum([g.shape[0] for g in gts])
        key = tuple(int(v) for v in offs)
        cache = getattr(self, "_gt_off_cache", None)
        if cache is None or cache[0] != (key, str(dev)):
            cache = self._gt_off_cache = ((key, str(dev)), torch.from_numpy(offs).to(dev))
        gt_cat = torch.cat(gts) if len(gts) > 1 else gts[0]
        lab_cat = torch.cat([l.to(dev).reshape(-1) for l in gt_labels]).to(torch.int32) if len(gt_labels) > 1 \
            else gt_labels[0].to(dev).reshape(-1).to(torch.int32)
        anchors = self.anchor_generator.grid_anchors([cls_score.shape[-2:]], device=dev)[0]
        cw = self.train_cfg.get("code_weight", None) or [1.0] * self.box_code_size
        l_cls, l_box, l_dir, _info = ops.anchor_loss(
            cls_score, bbox_pred, dir_pred, anchors.contiguous(), gt_cat.contiguous(), lab_cat.contiguous(), cache[1], self.num_classes,
            self.box_code_size, self.num_anchors, asg.pos_iou_thr, asg.neg_iou_thr, asg.min_pos_iou, self.loss_cls.gamma, self.loss_cls.alpha,
            self.loss_bbox.beta, self.dir_offset, self.diff_rad_by_sin, cw,
            (self.loss_cls.loss_weight, self.loss_bbox.loss_weight, self.loss_dir.loss_weight))
        return dict(loss_cls=[l_cls], loss_bbox=[l_box], loss_dir=[l_dir])

    def loss_from_targets(self, cls_score, bbox_pred, dir_pred, labels, label_weights, bbox_targets, bbox_weights,
                          dir_targets, dir_weights, num_total_samples):
        """``loss_single`` of the vendored head (det_anchor3d_head.py:192-277) for one level: (B, A*K, H, W) maps and
        flattened targets in, the three losses out."""
        labels, label_weights = labels.reshape(-1), label_weights.reshape(-1)
        bbox_targets = bbox_targets.reshape(-1, self.box_code_size)
        bbox_weights = bbox_weights.reshape(-1, self.box_code_size)
        dir_targets, dir_weights = dir_targets.reshape(-1), dir_weights.reshape(-1)
        cls_score = cls_score.permute(0, 2, 3, 1).reshape(-1, self.num_classes)
        loss_cls = self.loss_cls(cls_score, labels, label_weights, avg_factor=num_total_samples)
        # Regression / direction losses over ALL anchors with zero weight off the positives (the vendored head
        # gathers the positives first, det_anchor3d_head.py:236-262: same sums, but a gather needs a host sync).
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(-1, self.box_code_size)
        dir_pred = dir_pred.permute(0, 2, 3, 1).reshape(-1, 2)
        cw = self.train_cfg.get("code_weight", None)
        if cw:
            if getattr(self, "_code_weight", None) is None or self._code_weight.device != bbox_weights.device:
                self._code_weight = torch.tensor(cw, dtype=bbox_weights.dtype, device=bbox_weights.device)
            bbox_weights = bbox_weights * self._code_weight
        if self.diff_rad_by_sin:
            bbox_pred, bbox_targets = self.add_sin_difference(bbox_pred, bbox_targets)
        loss_bbox = self.loss_bbox(bbox_pred, bbox_targets, bbox_weights, avg_factor=num_total_samples)
        loss_dir = self.loss_dir(dir_pred, dir_targets, dir_weights, avg_factor=num_total_samples)
        return dict(loss_cls=[loss_cls], loss_bbox=[loss_bbox], loss_dir=[loss_dir])

    # ---- test time --------------------------------------------------------------------------
    @torch.no_grad()
    def get_bboxes(self, cls_scores, bbox_preds, dir_cls_preds, input_metas, cfg=None, rescale=False):
        """Decode + per-class rotated NMS per sample (vendored twin: det_anchor3d_head.py:374-423).
        Returns a list of (LiDARInstance3DBoxes, scores, labels) per sample."""
        assert len(cls_scores) == len(bbox_preds) == len(dir_cls_preds)
        sizes = [c.shape[-2:] for c in cls_scores]
        anchors = [a.reshape(-1, self.box_code_size)
                   for a in self.anchor_generator.grid_anchors(sizes, device=cls_scores[0].device)]
        return [self.get_bboxes_single([c[b].detach() for c in cls_scores], [r[b].detach() for r in bbox_preds],
                                       [d[b].detach() for d in dir_cls_preds], anchors, input_metas[b], cfg, rescale)
                for b in range(len(input_metas))]

    def get_bboxes_single(self, cls_scores, bbox_preds, dir_cls_preds, mlvl_anchors, input_meta, cfg=None,
                          rescale=False):
        """det_anchor3d_head.py:425-516: sigmoid scores, top-``nms_pre`` anchors by best class score per
        level, box decoding, background slot, ``box3d_multiclass_nms``, direction-bin yaw fix-up."""
        from .boxes import LiDARInstance3DBoxes, box3d_multiclass_nms, xywhr2xyxyr
        cfg = self.test_cfg if cfg is None else cfg
        box_type = input_meta.get("box_type_3d", LiDARInstance3DBoxes) if isinstance(input_meta, dict) \
            else LiDARInstance3DBoxes
        boxes, scores, dirs = [], [], []
        for cls_score, bbox_pred, dir_pred, anchors in zip(cls_scores, bbox_preds, dir_cls_preds, mlvl_anchors):
            assert cls_score.shape[-2:] == bbox_pred.shape[-2:] == dir_pred.shape[-2:]
            dir_bin = dir_pred.float().permute(1, 2, 0).reshape(-1, 2).max(dim=-1)[1]
            logits = cls_score.float().permute(1, 2, 0).reshape(-1, self.num_classes)
            sc = logits.sigmoid() if self.use_sigmoid_cls else logits.softmax(-1)
            deltas = bbox_pred.float().permute(1, 2, 0).reshape(-1, self.box_code_size)
            nms_pre = cfg.get("nms_pre", -1)
            if 0 < nms_pre < sc.shape[0]:
                best = sc.max(dim=1)[0] if self.use_sigmoid_cls else sc[:, :-1].max(dim=1)[0]
                top = best.topk(nms_pre)[1]
                anchors, deltas, sc, dir_bin = anchors[top], deltas[top], sc[top], dir_bin[top]
            boxes.append(self.bbox_coder.decode(anchors, deltas))
            scores.append(sc)
            dirs.append(dir_bin)
        boxes, scores, dirs = torch.cat(boxes), torch.cat(scores), torch.cat(dirs)
        for_nms = xywhr2xyxyr(box_type(boxes, box_dim=self.box_code_size).bev)
        if self.use_sigmoid_cls:
            scores = torch.cat([scores, scores.new_zeros(scores.shape[0], 1)], dim=1)
        boxes, scores, labels, dirs = box3d_multiclass_nms(boxes, for_nms, scores, cfg.get("score_thr", 0),
                                                           cfg["max_num"], cfg, dirs)
        if boxes.shape[0] > 0:
            rot = limit_period(boxes[..., 6] - self.dir_offset, self.dir_limit_offset, np.pi)
            boxes[..., 6] = rot + self.dir_offset + np.pi * dirs.to(boxes.dtype)
        return box_type(boxes, box_dim=self.box_code_size), scores, labels


# The reference vendors a copy of the upstream head under this name for its multi-task config
# (projects/mmdet3d_plugin/bevfusion/dense_heads/det_anchor3d_head.py:18); same arguments, same arithmetic.
HEADS.register_module(name="Anchor3DHeadV1", module=Anchor3DHead)
