"""``HardVFE`` / ``VFELayer`` of mmdet3d v0.17.1 (mmdet3d/models/voxel_encoders/voxel_encoder.py and
utils.py; un-vendored upstream pinned at /root/reference/README.md:153-156).  The reference reaches
them through ``pts_voxel_encoder=dict(type='HardVFE', in_channels=4, feat_channels=[64, 64],
with_cluster_center=True, with_voxel_center=True, ...)`` in
projects/configs/PointPillars_NewScenes/pointpillars_LiDAR.py:29-38 — the LiDAR stream of the
triple-modal stretch configuration.

PARITY UNPINNED: upstream is absent from the image and the reference holds no fixture for it; the
layer is restated from the published source and checked by hand-computed cases
(tests/test_pillars_cpu.py).  Upstream behaviour kept on purpose: padded point slots are zeroed
BEFORE the first layer only, so after Linear(no bias) -> BN -> ReLU they carry relu(beta - mean *
scale) and take part in the max, exactly as upstream.

State-dict names are upstream's (``vfe_layers.{i}.linear.weight``, ``vfe_layers.{i}.norm.*``).
Not built: ``fusion_layer`` (point-image fusion of MVX-Net) and ``with_distance`` — no NewScenes
config sets them."""
import torch
from torch import nn
from torch.nn import functional as F

from .bricks import build_norm_layer
from .registry import VOXEL_ENCODERS

__all__ = ["HardVFE", "VFELayer"]


class VFELayer(nn.Module):
    """Linear (no bias) -> BN over channels -> ReLU, then per voxel: nothing (``max_out=False``), the max over
    the point slots (``cat_max=False``) or [pointwise, max repeated] (``cat_max=True``)."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), max_out=True,
                 cat_max=True):
        super().__init__()
        self.cat_max, self.max_out = cat_max, max_out
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, inputs):
        k, t, _ = inputs.shape
        x = self.linear(inputs)
        pointwise = F.relu(self.norm(x.reshape(k * t, -1)).view(k, t, -1))     # BN over channels of (K*T, C)
        if not self.max_out:
            return pointwise
        aggregated = torch.max(pointwise, dim=1, keepdim=True)[0]
        if not self.cat_max:
            return aggregated.squeeze(1)
        return torch.cat([pointwise, aggregated.expand(-1, t, -1)], dim=2)


@VOXEL_ENCODERS.register_module()
class HardVFE(nn.Module):
    """(M, T, C) padded voxels + point counts + (M, 4) [batch, z, y, x] coordinates -> (M, feat_channels[-1])."""

    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), mode="max", fusion_layer=None,
                 return_point_feats=False):
        super().__init__()
        assert len(feat_channels) > 0
        if with_distance or fusion_layer is not None or return_point_feats:
            raise NotImplementedError("HardVFE: with_distance / fusion_layer / return_point_feats are used by no "
                                      "NewScenes config and are not built")
        in_channels += 3 * int(with_cluster_center) + 3 * int(with_voxel_center)
        self.in_channels = in_channels
        self._with_cluster_center, self._with_voxel_center = with_cluster_center, with_voxel_center
        self.vx, self.vy, self.vz = voxel_size[0], voxel_size[1], voxel_size[2]
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.point_cloud_range = point_cloud_range
        chans = [in_channels] + list(feat_channels)
        layers = []
        for i in range(len(chans) - 1):
            last = i == len(chans) - 2
            layers.append(VFELayer(chans[i] * (2 if i > 0 else 1), chans[i + 1], norm_cfg=norm_cfg, max_out=True,
                                   cat_max=not last))
        self.vfe_layers = nn.ModuleList(layers)
        self.num_vfe = len(layers)
        self.fusion_layer = None

    def forward(self, features, num_points, coors, img_feats=None, img_metas=None):
        parts = [features]
        if self._with_cluster_center:
            mean = features[:, :, :3].sum(dim=1, keepdim=True) / num_points.type_as(features).view(-1, 1, 1)
            parts.append(features[:, :, :3] - mean)
        if self._with_voxel_center:
            c = coors.type_as(features)
            centre = torch.stack((c[:, 3] * self.vx + self.x_offset, c[:, 2] * self.vy + self.y_offset,
                                  c[:, 1] * self.vz + self.z_offset), dim=-1)
            parts.append(features[:, :, :3] - centre.unsqueeze(1))
        x = torch.cat(parts, dim=-1)
        slots = torch.arange(x.shape[1], dtype=torch.int, device=x.device).view(1, -1)
        x = x * (num_points.int().unsqueeze(1) > slots).unsqueeze(-1).type_as(x)
        for vfe in self.vfe_layers:
            x = vfe(x)
        return x
