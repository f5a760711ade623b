"""SECOND backbone, SECONDFPN neck, PointPillarsScatter and the Voxelization layer with the
mmdet3d v0.17.1 names/arguments the reference configs use (bevfusion.py:46-74).  The upstream
package is not vendored in the reference; structure restated from its documented behaviour
 — parity unpinned by the reference, pinned here by shape/KAT tests.

Voxelization and PointPillarsScatter are the two radar-side subsystems north_star replaces: they
call the hand-written HIP kernels (omnihd_voxelize_hard, omnihd_pillar_scatter)."""
import torch
from torch import nn

from .. import ops
from .bricks import build_norm_layer, run_fused
from .registry import BACKBONES, MIDDLE_ENCODERS, NECKS


class Voxelization(nn.Module):
    """``pts_voxel_layer``: hard voxelisation of ONE point cloud (N,F) -> voxels, coors(z,y,x), num."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, deterministic=True):
        super().__init__()
        self.voxel_size, self.point_cloud_range = list(voxel_size), list(point_cloud_range)
        self.max_num_points = max_num_points
        self.max_voxels = tuple(max_voxels) if isinstance(max_voxels, (tuple, list)) else (max_voxels, max_voxels)

    def forward(self, points):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        return ops.hard_voxelize(points.contiguous().float(), self.voxel_size, self.point_cloud_range,
                                 self.max_num_points, max_voxels)

    def begin(self, points):
        """Enqueue the voxelisation and return a handle whose ``get()`` yields ``forward``'s result (device tensors
        only; CPU tensors — tests over the oracle — take the synchronous path)."""
        if not points.is_cuda:
            done = self.forward(points)
            return type("Done", (), {"get": staticmethod(lambda: done)})()
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        return ops.hard_voxelize_async(points.contiguous().float(), self.voxel_size, self.point_cloud_range,
                                       self.max_num_points, max_voxels)


@MIDDLE_ENCODERS.register_module()
class PointPillarsScatter(nn.Module):
    def __init__(self, in_channels, output_shape, channels_last=False):
        super().__init__()
        self.in_channels = in_channels
        self.ny, self.nx = output_shape
        self.channels_last = channels_last

    def forward(self, voxel_features, coors, batch_size=None):
        if batch_size is None:
            batch_size = int(coors[-1, 0]) + 1
        return ops.pillar_scatter(voxel_features, coors, int(batch_size), self.ny, self.nx, self.channels_last)


@BACKBONES.register_module()
class SECOND(nn.Module):
    def __init__(self, in_channels=128, out_channels=(128, 128, 256), layer_nums=(3, 5, 5), layer_strides=(2, 2, 2),
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv2d", bias=False), **_):
        super().__init__()
        in_filters = [in_channels, *out_channels[:-1]]
        blocks = []
        for i, n in enumerate(layer_nums):
            block = [nn.Conv2d(in_filters[i], out_channels[i], 3, stride=layer_strides[i], padding=1, bias=False),
                     build_norm_layer(norm_cfg, out_channels[i])[1], nn.ReLU(inplace=True)]
            for _ in range(n):
                block += [nn.Conv2d(out_channels[i], out_channels[i], 3, padding=1, bias=False),
                          build_norm_layer(norm_cfg, out_channels[i])[1], nn.ReLU(inplace=True)]
            blocks.append(nn.Sequential(*block))
        self.blocks = nn.ModuleList(blocks)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        outs = []
        for b in self.blocks:
            x = run_fused(b, x)
            outs.append(x)
        return tuple(outs)


@NECKS.register_module()
class SECONDFPN(nn.Module):
    def __init__(self, in_channels=(128, 128, 256), out_channels=(256, 256, 256), upsample_strides=(1, 2, 4),
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), upsample_cfg=dict(type="deconv", bias=False),
                 conv_cfg=dict(type="Conv2d", bias=False), use_conv_for_no_stride=False, **_):
        super().__init__()
        blocks = []
        for i, oc in enumerate(out_channels):
            s = upsample_strides[i]
            if s > 1 or (s == 1 and not use_conv_for_no_stride):
                up = nn.ConvTranspose2d(in_channels[i], oc, kernel_size=s, stride=s, bias=False)
            else:
                k = int(round(1 / s))
                up = nn.Conv2d(in_channels[i], oc, kernel_size=k, stride=k, bias=False)
            blocks.append(nn.Sequential(up, build_norm_layer(norm_cfg, oc)[1], nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(blocks)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        ups = [run_fused(d, x[i]) for i, d in enumerate(self.deblocks)]
        return [torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]]
