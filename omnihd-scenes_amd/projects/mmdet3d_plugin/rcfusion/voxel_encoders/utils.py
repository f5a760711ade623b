"""Pillar feature layers — mirror of projects/mmdet3d_plugin/rcfusion/voxel_encoders/utils.py
(``get_paddings_indicator`` :9-29, ``PFNLayer`` :107-181, ``PFNLayer_Radar`` :183-280): same class
names, constructor arguments, parameter names (``linear``, ``norm`` / ``linear{1,2,3}``,
``norm{1,2,3}``) and outputs.  The normalisation is applied on the (N*M, C) view of the point
features — numerically the same as the reference's permute -> BatchNorm1d -> permute pair
(utils.py:161-162), without the two transposed copies."""
import torch
from torch import nn
from torch.nn import functional as F

from omnihd_amd.mm import build_norm_layer

__all__ = ["get_paddings_indicator", "PFNLayer", "PFNLayer_Radar"]


def get_paddings_indicator(actual_num, max_num, axis=0):
    """(N,) point counts -> (N, max_num) bool mask of the valid slots."""
    actual_num = torch.unsqueeze(actual_num, axis + 1)
    shape = [1] * actual_num.dim()
    shape[axis + 1] = -1
    slots = torch.arange(max_num, dtype=torch.int, device=actual_num.device).view(shape)
    return actual_num.int() > slots


def _norm_points(norm, x):
    """BatchNorm over the channel dim of (N, M, C) point features."""
    n, m, c = x.shape
    return norm(x.reshape(n * m, c)).view(n, m, c)


def _pool(x, mode, num_voxels, aligned_distance):
    if aligned_distance is not None:
        x = x.mul(aligned_distance.unsqueeze(-1))
    if mode == "max":
        return x, torch.max(x, dim=1, keepdim=True)[0]
    return x, x.sum(dim=1, keepdim=True) / num_voxels.type_as(x).view(-1, 1, 1)


class PFNLayer(nn.Module):
    """Linear (no bias) -> BN over channels -> ReLU -> max/avg over the points of a pillar."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01),
                 last_layer=False, mode="max"):
        super().__init__()
        self.name = "PFNLayer"
        self.last_vfe = last_layer
        if not self.last_vfe:
            out_channels = out_channels // 2
        self.units = out_channels
        self.norm = build_norm_layer(norm_cfg, self.units)[1]
        self.linear = nn.Linear(in_channels, self.units, bias=False)
        assert mode in ["max", "avg"]
        self.mode = mode

    def forward(self, inputs, num_voxels=None, aligned_distance=None):
        x = F.relu(_norm_points(self.norm, self.linear(inputs)))
        x, x_max = _pool(x, self.mode, num_voxels, aligned_distance)
        if self.last_vfe:
            return x_max
        return torch.cat([x, x_max.repeat(1, inputs.shape[1], 1)], dim=2)


class PFNLayer_Radar(nn.Module):
    """RCFusion's three-branch layer: spatial / velocity / power-SNR channel groups each get their
    own Linear+BN (units split 1/2, 1/4, 1/4), concatenated before ReLU (reference :183-280)."""

    SPATIAL = (0, 1, 2, 7, 8, 9, 10, 11)
    VELOCITY = (3, 4, 12, 13)
    SNR = (5, 6, 14, 15)

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01),
                 last_layer=False, mode="max"):
        super().__init__()
        self.name = "PFNLayer"
        self.last_vfe = last_layer
        self.in_channels1, self.in_channels2, self.in_channels3 = 8, 4, 4
        if not self.last_vfe:
            out_channels = out_channels // 2
        self.units1, self.units2, self.units3 = out_channels // 2, out_channels // 4, out_channels // 4
        self.norm1 = build_norm_layer(norm_cfg, self.units1)[1]
        self.norm2 = build_norm_layer(norm_cfg, self.units2)[1]
        self.norm3 = build_norm_layer(norm_cfg, self.units3)[1]
        self.linear1 = nn.Linear(self.in_channels1, self.units1, bias=False)
        self.linear2 = nn.Linear(self.in_channels2, self.units2, bias=False)
        self.linear3 = nn.Linear(self.in_channels3, self.units3, bias=False)
        assert mode in ["max", "avg"]
        self.mode = mode

    def forward(self, inputs, num_voxels=None, aligned_distance=None):
        dev = inputs.device
        parts = []
        for idx, lin, norm in ((self.SPATIAL, self.linear1, self.norm1), (self.VELOCITY, self.linear2, self.norm2),
                               (self.SNR, self.linear3, self.norm3)):
            sel = inputs.index_select(2, torch.tensor(idx, device=dev))
            parts.append(_norm_points(norm, lin(sel)))
        x = F.relu(torch.cat(parts, dim=-1))
        x, x_max = _pool(x, self.mode, num_voxels, aligned_distance)
        if self.last_vfe:
            return x_max
        return torch.cat([x, x_max.repeat(1, inputs.shape[1], 1)], dim=2)
