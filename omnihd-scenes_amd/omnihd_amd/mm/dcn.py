"""Deformable convolution v1 ("DCN" = mmcv DeformConv2dPack) on plain torch ops.

The reference builds it with ``build_conv_layer(dict(type='DCN', groups=4, im2col_step=128))``
inside DepthNet (bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:587-595); the op itself is
an mmcv extension that is not vendored.  It is NOT one of the subsystems north_star replaces, so it
is expressed with library ops: the bilinear sampling of all taps is ONE weighted row gather over the
channels-last feature rows (``embedding_bag`` with 4 corner rows per (pixel, tap) bag — its backward is
a sort + segmented sum, no atomics), followed by one dense GEMM with a block-diagonal (grouped) weight.
(A first version used nine ``grid_sample`` calls; their atomic-scatter backward alone cost 7.6 ms of a
70 ms training step.)  Semantics: offsets from a zero-initialised ``conv_offset`` (so the layer starts
as a plain grouped conv), zero padding outside the image, offset channel order
(deform_group, tap, (dy, dx)) as in mmcv.
"""
import os
from .._env import env as _env

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
        gemm_dtype = torch.bfloat16 if (torch.is_autocast_enabled() and x.is_cuda) else torch.float32
        if x.is_cuda and x.dtype in (torch.bfloat16, torch.float32):
            from .. import ops
            if ops.dcn3x3_supported(x, self.k, self.stride, self.deform_groups):
                with torch.autocast(x.device.type, enabled=False):
                    return self._hip_sample_and_gemm(x, offset, gemm_dtype)
        with torch.autocast(x.device.type, enabled=False):             # sampling positions need fp32
            if self.deform_groups == 1:
                return self._gather_and_gemm(x.float(), offset.float(), gemm_dtype).to(x.dtype)
            return self._sample_and_contract(x.float(), offset.float()).to(x.dtype)

    def _hip_sample_and_gemm(self, x, offset, gemm_dtype):
        """Training path on the GPU: hand-written HIP row-gather kernels (omnihd_dcn3x3_sample_*) + one GEMM, in bf16
        under autocast and in fp32 otherwise."""
        from .. import ops
        B, C, H, W = x.shape
        xb = x.to(gemm_dtype).permute(0, 2, 3, 1).contiguous()                   # (B,H,W,C): a view for channels-last x
        ob = offset.float().permute(0, 2, 3, 1).contiguous()                     # (B,Ho,Wo,18), channel = tap*2 + (dy,dx)
        col = ops.dcn3x3_sample(xb, ob, self.stride, self.padding, self.dilation)
        Ho, Wo = ob.shape[1:3]
        wmat = self._grouped_weight()                                            # (9*Cin, Cout) fp32, block-diagonal
        if gemm_dtype == torch.float32 and _env("OMNIHD_FP32_CONV", "tune") != "miopen":
            # the fp32 step: the contraction is a 1x1 convolution over the column rows — on the fp32-grade split kernels
            # (3-term bf16 MFMA, csrc/conv_igemm.hip) where they apply and measure faster than the fp32 GEMM library.  w4 is a
            # temporary of this forward: ops.split_weight does not cache the planes of non-leaf weights.
            rows = col.view(1, B * Ho * Wo, 1, col.shape[1]).permute(0, 3, 1, 2)   # (1, 9*Cin, M, 1) over (M, 9*Cin) memory
            w4 = wmat.t().reshape(self.out_channels, col.shape[1], 1, 1)
            if ops.conv_split_supported(rows, w4, (1, 1), (0, 0), (1, 1)):
                out = ops.conv_split(rows, w4, None, (1, 1), (0, 0), (1, 1))        # (1, Cout, M, 1) channels-last
                return out.permute(0, 2, 3, 1).reshape(B, Ho, Wo, self.out_channels).permute(0, 3, 1, 2).to(x.dtype)
        out = col @ wmat.to(gemm_dtype)
        return out.view(B, Ho, Wo, self.out_channels).permute(0, 3, 1, 2).to(x.dtype)

    def _grouped_weight(self):
        """(k*k*Cin, Cout) block-diagonal matrix of the grouped weight, row index = tap*Cin + c."""
        k, g = self.k, self.groups
        cg, og = self.in_channels // g, self.out_channels // g
        w = self.weight.float()
        full = w.new_zeros(k * k, self.in_channels, self.out_channels)
        for i in range(g):
            full[:, i * cg:(i + 1) * cg, i * og:(i + 1) * og] = w[i * og:(i + 1) * og].permute(2, 3, 1, 0).reshape(k * k, cg, og)
        return full.reshape(k * k * self.in_channels, self.out_channels)

    def _gather_and_gemm(self, x, offset, gemm_dtype):
        B, C, H, W = x.shape
        k, s, p, d = self.k, self.stride, self.padding, self.dilation
        Ho, Wo = offset.shape[-2:]
        off = offset.view(B, k * k, 2, Ho, Wo)
        ky = torch.arange(k, device=x.device, dtype=x.dtype).repeat_interleave(k).view(1, k * k, 1, 1) * d
        kx = torch.arange(k, device=x.device, dtype=x.dtype).repeat(k).view(1, k * k, 1, 1) * d
        ys = torch.arange(Ho, device=x.device, dtype=x.dtype).view(1, 1, Ho, 1) * s - p
        xs = torch.arange(Wo, device=x.device, dtype=x.dtype).view(1, 1, 1, Wo) * s - p
        py = (ys + ky + off[:, :, 0]).permute(0, 2, 3, 1)              # (B, Ho, Wo, taps)
        px = (xs + kx + off[:, :, 1]).permute(0, 2, 3, 1)
        y0, x0 = torch.floor(py), torch.floor(px)
        fy, fx = py - y0, px - x0
        y0, x0 = y0.long(), x0.long()
        base = (torch.arange(B, device=x.device) * (H * W)).view(B, 1, 1, 1)
        # mmcv's sampling gate (deformable_im2col: `h_im > -1 && w_im > -1 && h_im < height && w_im < width`): a tap whose position
        # lies ON or beyond that border contributes nothing — and, unlike a per-corner mask, has NO offset gradient either (at
        # the zero-initialised offsets every border tap sits exactly on -1: the HIP kernels and oracle/dcn_oracle.c gate it)
        gate = (py > -1) & (px > -1) & (py < H) & (px < W)
        idx, wts = [], []
        for dy_, wy in ((0, 1 - fy), (1, fy)):
            for dx_, wx in ((0, 1 - fx), (1, fx)):
                yy, xx = y0 + dy_, x0 + dx_
                ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W) & gate    # zero padding outside the image
                idx.append(torch.where(ok, base + yy * W + xx, torch.zeros_like(yy)))
                wts.append(wy * wx * ok)
        idx = torch.stack(idx, dim=-1).reshape(-1, 4)                   # one bag of 4 corner rows per (pixel, tap)
        wts = torch.stack(wts, dim=-1).reshape(-1, 4)
        rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
        col = F.embedding_bag(idx, rows, per_sample_weights=wts, mode="sum")      # (B*Ho*Wo*taps, C)
        col = col.view(B * Ho * Wo, k * k * C).to(gemm_dtype)
        out = col @ self._grouped_weight().to(gemm_dtype)
        return out.view(B, Ho, Wo, self.out_channels).permute(0, 3, 1, 2)

    def _sample_and_contract(self, x, offset):
        """Reference formulation with one ``grid_sample`` per tap (kept for deform_groups > 1 and as the test oracle
        of :meth:`_gather_and_gemm`)."""
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
