"""Pillar feature nets — mirror of projects/mmdet3d_plugin/rcfusion/voxel_encoders/pillar_encoder.py
(``RadarPillarFeatureNet`` :11-153, ``PillarFeatureNetV1`` :302-432).  Registered under the same
``type=`` names; same constructor arguments; same decoration of the raw pillar points:
cluster offset (xyz - mean), pillar-centre offset and — radar variant — velocity/SNR offsets.

``legacy=True`` (the default the configs use) reproduces the reference's in-place behaviour: the
pillar-centre offset OVERWRITES the raw x,y channels as well (pillar_encoder.py:410-416), so the
decorated tensor starts with (x-cx, y-cy, z, ...).  The caller's ``features`` tensor is not modified
here (the reference mutates it; nothing reads it afterwards)."""
import torch
from torch import nn

from omnihd_amd.mm import VOXEL_ENCODERS

from .utils import PFNLayer, PFNLayer_Radar, get_paddings_indicator

__all__ = ["PillarFeatureNetV1", "RadarPillarFeatureNet"]


class _PillarNetBase(nn.Module):
    layer_cls = PFNLayer

    def _setup(self, in_channels, feat_channels, with_distance, with_cluster_center, with_voxel_center, voxel_size,
               point_cloud_range, norm_cfg, mode, legacy, extra=0):
        assert len(feat_channels) > 0
        self.legacy = legacy
        in_channels += 3 * bool(with_cluster_center) + 2 * bool(with_voxel_center) + bool(with_distance) + extra
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.in_channels = in_channels
        chans = [in_channels] + list(feat_channels)
        self.pfn_layers = nn.ModuleList([
            self.layer_cls(chans[i], chans[i + 1], norm_cfg=norm_cfg, last_layer=(i == len(chans) - 2), mode=mode)
            for i in range(len(chans) - 1)])
        self.vx, self.vy = voxel_size[0], voxel_size[1]
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.point_cloud_range = point_cloud_range

    def _decorate(self, features, num_points, coors):
        features = features.float()
        cnt = num_points.type_as(features).view(-1, 1, 1)
        parts = [None]
        if self._with_cluster_center:
            parts.append(features[:, :, :3] - features[:, :, :3].sum(dim=1, keepdim=True) / cnt)
        base = features
        if self._with_voxel_center:
            cx = coors[:, 3].type_as(features).unsqueeze(1) * self.vx + self.x_offset
            cy = coors[:, 2].type_as(features).unsqueeze(1) * self.vy + self.y_offset
            f_center = torch.stack([features[:, :, 0] - cx, features[:, :, 1] - cy], dim=-1)
            if self.legacy:   # the reference's f_center is a VIEW of features[:, :, :2]: raw x,y are replaced
                base = torch.cat([f_center, features[:, :, 2:]], dim=-1)
            parts.append(f_center)
        if self._with_distance:
            parts.append(torch.norm(base[:, :, :3], 2, 2, keepdim=True))
        parts[0] = base
        return parts, cnt, base

    def _run(self, parts, num_points):
        features = torch.cat(parts, dim=-1)
        mask = get_paddings_indicator(num_points, features.shape[1], axis=0).unsqueeze(-1).type_as(features)
        features = features * mask
        for pfn in self.pfn_layers:
            features = pfn(features, num_points)
        return features.squeeze()


@VOXEL_ENCODERS.register_module()
class PillarFeatureNetV1(_PillarNetBase):
    def __init__(self, in_channels=4, feat_channels=(64, ), with_distance=False, with_cluster_center=True,
                 with_voxel_center=True, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), mode="max", legacy=True):
        super().__init__()
        self._setup(in_channels, feat_channels, with_distance, with_cluster_center, with_voxel_center, voxel_size,
                    point_cloud_range, norm_cfg, mode, legacy)

    def forward(self, features, num_points, coors, img_feats=None, img_metas=None):
        parts, _, _ = self._decorate(features, num_points, coors)
        return self._run(parts, num_points)


@VOXEL_ENCODERS.register_module()
class RadarPillarFeatureNet(_PillarNetBase):
    layer_cls = PFNLayer_Radar

    def __init__(self, in_channels=7, feat_channels=(64, ), with_distance=False, with_cluster_center=True,
                 with_voxel_center=True, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), mode="max", legacy=True,
                 with_velocity_snr_center=True):
        super().__init__()
        self._with_velocity_snr_center = with_velocity_snr_center
        self._setup(in_channels, feat_channels, with_distance, with_cluster_center, with_voxel_center, voxel_size,
                    point_cloud_range, norm_cfg, mode, legacy, extra=4 * bool(with_velocity_snr_center))

    def forward(self, features, num_points, coors, img_feats=None, img_metas=None):
        parts, cnt, base = self._decorate(features, num_points, coors)
        if self._with_velocity_snr_center:
            parts.append(base[:, :, 3:7] - base[:, :, 3:7].sum(dim=1, keepdim=True) / cnt)
        return self._run(parts, num_points)
