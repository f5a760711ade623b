"""Occupancy head of the multi-task fusion config — mirror of the reference's
``BEVOCCHead2Dv2`` (projects/mmdet3d_plugin/bevfusion/dense_heads/bev_occ_head.py:719-895): a 3x3 ConvModule on
the fused BEV feature, a per-cell MLP (Linear, Softplus, Linear) that expands each BEV cell into Dz x n_cls
logits, cross-entropy + the two scene-completion "scal" losses.

The reference's ``sem_scal_loss`` loops over the classes with three ``if torch.sum(...) > 0`` host reads each
(36 device synchronisations per step); here the per-class sums are one masked reduction and the conditions are
masks — same value, no synchronisation."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from omnihd_amd.mm import ConvModule
from omnihd_amd.mm.registry import HEADS, LOSSES, build_from_cfg


def _nll_of_one(x):
    """``F.binary_cross_entropy(x, ones)`` = -log(x) with torch's clamp of the log at -100, written as a clamp of
    the argument at e^-100 so that the gradient is finite (zero) below it."""
    return -torch.log(x.clamp_min(3.7200759760208e-44))


@HEADS.register_module()
class BEVOCCHead2Dv2(nn.Module):
    def __init__(self, in_dim=256, out_dim=256, Dz=16, use_mask=False, num_classes=19, use_predicter=True,
                 class_balance=False, loss_occ=None):
        super().__init__()
        self.in_dim, self.out_dim, self.Dz = in_dim, out_dim, Dz
        out_channels = out_dim if use_predicter else num_classes * Dz
        self.final_conv = ConvModule(in_dim, out_channels, kernel_size=3, stride=1, padding=1, bias=True,
                                     conv_cfg=dict(type="Conv2d"))
        self.use_predicter = use_predicter
        if use_predicter:
            self.predicter = nn.Sequential(nn.Linear(out_dim, out_dim * 2), nn.Softplus(),
                                           nn.Linear(out_dim * 2, num_classes * Dz))
        self.use_mask, self.num_classes, self.class_balance = use_mask, num_classes, class_balance
        self.loss_occ = build_from_cfg(loss_occ, LOSSES)

    def init_weights(self):
        pass

    def forward(self, occ_feats):
        """[(B, C, Dy, Dx)] -> (B, Dx, Dy, Dz, n_cls) logits."""
        occ_pred = self.final_conv(occ_feats[0]).permute(0, 3, 2, 1)
        bs, Dx, Dy = occ_pred.shape[:3]
        if self.use_predicter:
            occ_pred = self.predicter(occ_pred).view(bs, Dx, Dy, self.Dz, self.num_classes)
        return occ_pred

    def loss(self, occ_pred, gt_occ):
        sem = gt_occ.long()
        occ_pred = occ_pred.float()
        loss_ssc = self.sem_scal_loss(occ_pred, sem) + self.geo_scal_loss(occ_pred, sem)
        loss_occ = self.loss_occ(occ_pred.reshape(-1, self.num_classes), sem.reshape(-1))
        return dict(loss_ssc=loss_ssc, loss_occ=loss_occ)

    @staticmethod
    def geo_scal_loss(preds, ssc_target, semantic=True):
        """Precision / recall / specificity of "occupied" (class != 0) over the known voxels (target != 255)."""
        if semantic:
            empty = F.softmax(preds, dim=-1)[..., 0]
        else:
            empty = 1 - torch.sigmoid(preds[..., 0])
        known = (ssc_target != 255).to(empty.dtype)
        occupied_t = ((ssc_target != 0) & (ssc_target != 255)).to(empty.dtype)
        free_t = known - occupied_t
        occupied_p = (1 - empty) * known
        inter = (occupied_t * occupied_p).sum()
        precision = inter / occupied_p.sum()
        recall = inter / occupied_t.sum()
        spec = (free_t * empty).sum() / free_t.sum()
        return _nll_of_one(precision) + _nll_of_one(recall) + _nll_of_one(spec)

    @staticmethod
    def sem_scal_loss(preds, ssc_target):
        """Mean over the classes present in the target of -log precision - log recall - log specificity."""
        p = F.softmax(preds, dim=-1)
        n_cls = p.shape[-1]
        known = ssc_target != 255
        p = p * known.unsqueeze(-1)
        onehot = (ssc_target.unsqueeze(-1) == torch.arange(n_cls, device=p.device)) & known.unsqueeze(-1)
        onehot = onehot.to(p.dtype)
        flat_p, flat_t = p.reshape(-1, n_cls), onehot.reshape(-1, n_cls)
        n_known = known.sum().to(p.dtype)
        nominator = (flat_p * flat_t).sum(0)
        sum_p, sum_t = flat_p.sum(0), flat_t.sum(0)
        present = sum_t > 0


        def term(num, den, cond):
            """-log(num / den) where ``cond`` holds, 0 elsewhere — the ratio is forced to 1 there BEFORE the log, so
            neither the value nor the gradient of a skipped term is ever inf/nan."""
            ratio = num / torch.where(den > 0, den, torch.ones_like(den))
            return _nll_of_one(torch.where(cond, ratio, torch.ones_like(ratio)))
        n_other = n_known - sum_t
        loss_c = term(nominator, sum_p, present & (sum_p > 0)) + term(nominator, sum_t, present)
        loss_c = loss_c + term(n_known - sum_p - sum_t + nominator, n_other, present & (n_other > 0))   # sum (1-p)(1-t)
        return loss_c.sum() / present.sum()

    def get_occ(self, occ_pred, img_metas=None):
        """(B, Dx, Dy, Dz, n_cls) -> list of (Dx, Dy, Dz) uint8 class maps."""
        return list(occ_pred.softmax(-1).argmax(-1).cpu().numpy().astype(np.uint8))
