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
config sets them.

PACKED EVALUATION (``packed=True`` / ``OMNIHD_VFE_PACKED=1``; off by default until its first GPU run).
The dense formulation above pushes every one of the M x T slots through every layer: at the LiDAR
stream's sizes (30 000-40 000 pillars x 64 slots per sample, ~3 points per pillar on average) that
is 15 M rows x 64..128 channels per flat batch of six samples — several multi-GB passes, ~95 % of them
over empty slots.  But an empty slot's value is the same for every empty slot of a voxel (layer 1: the
constant relu(bn(0)); deeper layers: a function of the voxel's aggregate only), so the same numbers
follow from the REAL points plus ONE representative row per voxel that stands for its (T - n) empty
slots: the representative takes part in the max when T - n > 0 and enters the BatchNorm statistics with
weight T - n.  Work and traffic then scale with the number of points, not slots (20-60x less here),
with identical outputs, running statistics and gradients (tests/test_pillars_cpu.py).  Real points are
packed without any host synchronisation into a buffer of the stream's input size (a cumulative sum
over the slot mask gives every real slot its place)."""
from .._env import env as _env
import os

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
                 return_point_feats=False, packed=None):
        super().__init__()
        self.packed = packed          # None: follow OMNIHD_VFE_PACKED at call time (default 0)
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

    def forward(self, features, num_points, coors, img_feats=None, img_metas=None, max_real_points=None):
        packed = self.packed if self.packed is not None else _env("OMNIHD_VFE_PACKED", "0") == "1"
        if packed:
            return self._forward_packed(features, num_points, coors, max_real_points)
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

    # ---- packed evaluation: real points + one representative row per voxel ---------------------------------------
    @staticmethod
    def _weighted_norm(norm, rows, w_rows, reps, w_reps, n_total):
        """BatchNorm of the M*T dense rows expressed on (rows with 0/1 weights) + (representatives with weights T-n):
        the statistics, the running-statistics update and the exchange between ranks are the layer's own."""
        from torch import distributed as dist
        if rows.dtype in (torch.bfloat16, torch.float16):      # statistics and normalisation in fp32 whatever autocast
            rows, reps = rows.float(), reps.float()            # made of the Linear
        if not norm.training:
            scale = norm.weight * torch.rsqrt(norm.running_var + norm.eps)
            shift = norm.bias - norm.running_mean * scale
            return rows * scale + shift, reps * scale + shift
        s1 = (rows * w_rows).sum(0) + (reps * w_reps).sum(0)
        s2 = (rows * rows * w_rows).sum(0) + (reps * reps * w_reps).sum(0)
        mean, meansqr = s1 / n_total, s2 / n_total
        synced = getattr(norm, "_omnihd_sync", False) and dist.is_available() and dist.is_initialized() \
            and dist.get_world_size() > 1
        if synced:          # naiveSyncBN: mean of the per-rank means (mm/sync_bn.py)
            from .sync_bn import AllReduceSum
            vec = AllReduceSum.apply(torch.cat([mean, meansqr])) * (1.0 / dist.get_world_size())
            mean, meansqr = torch.split(vec, mean.numel())
        var = meansqr - mean * mean
        with torch.no_grad():
            if norm.track_running_stats and norm.momentum is not None:
                unbias = 1.0 if synced else n_total / max(n_total - 1, 1)          # torch's BatchNorm keeps the unbiased one
                norm.running_mean += norm.momentum * (mean - norm.running_mean)
                norm.running_var += norm.momentum * (var * unbias - norm.running_var)
                if norm.num_batches_tracked is not None and not synced:      # as the layer's own two branches do
                    norm.num_batches_tracked += 1
        scale = norm.weight * torch.rsqrt(var + norm.eps)
        shift = norm.bias - mean * scale
        return rows * scale + shift, reps * scale + shift

    def _forward_packed(self, features, num_points, coors, max_real_points=None):
        M, T, C = features.shape
        dev = features.device
        n = num_points.to(torch.long).clamp(max=T)
        cap = int(max_real_points) if max_real_points is not None else M * T       # upper bound of real slots, host-known
        cap = min(cap, M * T)
        slots = torch.arange(T, device=dev).view(1, T)
        mask = (slots < n.view(M, 1)).reshape(-1)                                   # (M*T,) real slots
        # place of every real slot in the packed buffer; slots beyond ``cap`` cannot exist when cap >= #points in
        place = torch.cumsum(mask, 0) - 1
        src = torch.full((cap + 1,), M * T, dtype=torch.long, device=dev)           # M*T = a zero dummy row
        src.scatter_(0, torch.where(mask, place.clamp(max=cap), torch.full_like(place, cap)), torch.arange(M * T, device=dev))
        src = src[:cap]
        live = (src < M * T)
        w_rows = live.to(features.dtype).unsqueeze(1)
        vox = torch.where(live, src // T, torch.full_like(src, M))                  # dummy rows belong to voxel M
        flat = torch.cat([features.reshape(M * T, C), features.new_zeros(1, C)], 0)
        pts = flat[src]                                                             # (cap, C) real points, zeros elsewhere
        parts = [pts]
        if self._with_cluster_center:      # the dense expression: sum over ALL slots / count (empty slots are zero rows)
            mean = features[:, :, :3].sum(dim=1) / num_points.type_as(features).view(-1, 1)
            parts.append(pts[:, :3] - torch.cat([mean, mean.new_zeros(1, 3)], 0)[vox])
        if self._with_voxel_center:
            c = coors.type_as(features)
            centre = torch.stack((c[:, 3] * self.vx + self.x_offset, c[:, 2] * self.vy + self.y_offset,
                                  c[:, 1] * self.vz + self.z_offset), dim=-1)
            parts.append(pts[:, :3] - torch.cat([centre, centre.new_zeros(1, 3)], 0)[vox])
        rows = torch.cat(parts, dim=-1) * w_rows                                    # masked decorated points
        reps = rows.new_zeros(M, rows.shape[1])                                     # an empty slot's input is a zero row
        w_reps = (T - n).to(rows.dtype).unsqueeze(1)                                # how many empty slots it stands for
        has_empty = (n < T).unsqueeze(1)
        n_total = float(M * T)
        for i, vfe in enumerate(self.vfe_layers):
            yr, yp = self._weighted_norm(vfe.norm, vfe.linear(rows), w_rows, vfe.linear(reps), w_reps, n_total)
            pr, pp = F.relu(yr), F.relu(yp)
            agg = pr.new_full((M + 1, pr.shape[1]), float("-inf"))
            agg = agg.scatter_reduce(0, vox.unsqueeze(1).expand(-1, pr.shape[1]), pr, reduce="amax", include_self=True)[:M]
            agg = torch.where(has_empty, torch.maximum(agg, pp), agg)
            if not vfe.cat_max:
                return agg
            rows = torch.cat([pr, torch.cat([agg, agg.new_zeros(1, agg.shape[1])], 0)[vox]], dim=1) * w_rows
            reps = torch.cat([pp, agg], dim=1)
        return rows
