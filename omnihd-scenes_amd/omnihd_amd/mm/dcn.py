"""Deformable convolution v1 ("DCN" = mmcv DeformConv2dPack) on plain torch ops.

The reference builds it with ``build_conv_layer(dict(type='DCN', groups=4, im2col_step=128))``
inside DepthNet (bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:587-595); the op itself is
an mmcv extension that is not vendored.  It is NOT one of the subsystems north_star replaces, so it
is expressed here with bilinear ``grid_sample`` gathers + one grouped 1x1 contraction (MIOpen /
rocBLAS underneath).  Semantics: offsets from a zero-initialised ``conv_offset`` (so the layer starts
as a plain grouped conv), zero padding outside the image, offset channel order
(deform_group, tap, (dy, dx)) as in mmcv.
"""
import torch
import torch.nn.functional as F
from torch import nn


class DeformConv2dPack(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=False, im2col_step=32, **_unused):
        super().__init__()
        assert not bias, "mmcv DeformConv2d has no bias"
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        self.in_channels, self.out_channels, self.k = in_channels, out_channels, k
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deform_groups = groups, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, k, k))
        nn.init.kaiming_uniform_(self.weight, nonlinearity="relu")
        self.conv_offset = nn.Conv2d(in_channels, deform_groups * 2 * k * k, kernel_size=k, stride=stride,
                                     padding=padding, dilation=dilation, bias=True)
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x):
        offset = self.conv_offset(x)                                   # (B, dg*2*k*k, Ho, Wo)
        with torch.autocast(x.device.type, enabled=False):             # sampling positions need fp32
            return self._sample_and_contract(x.float(), offset.float()).to(x.dtype)

    def _sample_and_contract(self, x, offset):
        B, C, H, W = x.shape
        k, s, p, d = self.k, self.stride, self.padding, self.dilation
        Ho, Wo = offset.shape[-2:]
        dg = self.deform_groups
        offset = offset.view(B, dg, k * k, 2, Ho, Wo)
        ys = torch.arange(Ho, device=x.device, dtype=x.dtype).view(1, 1, Ho, 1) * s - p
        xs = torch.arange(Wo, device=x.device, dtype=x.dtype).view(1, 1, 1, Wo) * s - p
        xg = x.reshape(B * dg, C // dg, H, W)
        cols = []
        for t in range(k * k):
            ky, kx = divmod(t, k)
            py = ys + ky * d + offset[:, :, t, 0]                      # (B, dg, Ho, Wo)
            px = xs + kx * d + offset[:, :, t, 1]
            gx = px * (2.0 / max(W - 1, 1)) - 1.0
            gy = py * (2.0 / max(H - 1, 1)) - 1.0
            grid = torch.stack([gx, gy], dim=-1).view(B * dg, Ho, Wo, 2)
            cols.append(F.grid_sample(xg, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
                        .view(B, C, Ho, Wo))
        col = torch.stack(cols, dim=2)                                 # (B, C, k*k, Ho, Wo)
        g = self.groups
        col = col.reshape(B, g, (C // g) * k * k, Ho * Wo)
        w = self.weight.float().reshape(g, self.out_channels // g, (C // g) * k * k)
        out = torch.einsum("gok,bgkn->bgon", w, col)
        return out.reshape(B, self.out_channels, Ho, Wo)
