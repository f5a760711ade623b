"""BEVPoolv2 autograd operator with the reference's Python API.

Mirrors ``projects/mmdet3d_plugin/ops/bev_pool_v2/bev_pool.py`` of the reference (class
``QuickCumsumCuda`` :11-83, ``bev_pool_v2`` :86-92, ``TRTBEVPoolv2`` :95-142): same names, same
argument order and meaning, same outputs.  Differences that are invisible to callers:

* the kernels are HIP (gfx950) behind a C ABI;
* the backward tables (stable sort by ``ranks_feat`` + run lengths, reference :47-57) are built on
  the device and cached per ``ranks_*`` tensor identity, since they only change with calibration;
* ``bev_feat_shape`` entries may be python ints or 0-d tensors (the reference passes ``nx[i]``).
"""
import torch

from omnihd_amd import ops as _ops

from . import bev_pool_v2_ext

__all__ = ["bev_pool_v2", "TRTBEVPoolv2"]

_BP_CACHE = {}
_BP_CACHE_MAX = 8


def _backward_tables(ranks_bev, ranks_depth, ranks_feat, n_feat_rows):
    key = (ranks_feat.data_ptr(), ranks_feat._version, ranks_feat.numel(), ranks_bev.data_ptr(),
           ranks_bev._version, ranks_depth.data_ptr(), ranks_depth._version, str(ranks_feat.device))
    hit = _BP_CACHE.get(key)
    if hit is None:
        if len(_BP_CACHE) >= _BP_CACHE_MAX:
            _BP_CACHE.pop(next(iter(_BP_CACHE)))
        tables = _ops.backward_tables(ranks_bev, ranks_depth, ranks_feat, n_feat_rows)
        # keep the source tensors alive so their data_ptr cannot be recycled while cached
        hit = (tables, (ranks_bev, ranks_depth, ranks_feat))
        _BP_CACHE[key] = hit
    return hit[0]


class QuickCumsumCuda(torch.autograd.Function):
    """BEVPoolv2 (https://arxiv.org/abs/2211.17111): depth (B,N,D,H,W) x feat (B,N,H,W,C) pooled
    into a (B,Z,Y,X,C) buffer through the rank tables."""

    @staticmethod
    def forward(ctx, depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                interval_starts, interval_lengths):
        depth = depth.contiguous().float()
        feat = feat.contiguous().float()
        ranks_bev = ranks_bev.contiguous().int()
        ranks_depth = ranks_depth.contiguous().int()
        ranks_feat = ranks_feat.contiguous().int()
        interval_lengths = interval_lengths.contiguous().int()
        interval_starts = interval_starts.contiguous().int()
        shape = tuple(int(s) for s in bev_feat_shape)

        out = feat.new_zeros(shape)
        bev_pool_v2_ext.bev_pool_v2_forward(depth, feat, out, ranks_depth, ranks_feat, ranks_bev,
                                            interval_lengths, interval_starts)
        ctx.save_for_backward(ranks_bev, depth, feat, ranks_feat, ranks_depth)
        return out

    @staticmethod
    def backward(ctx, out_grad):
        ranks_bev, depth, feat, ranks_feat, ranks_depth = ctx.saved_tensors
        n_feat_rows = feat.numel() // feat.size(-1)
        rb, rd, rf, starts_bp, lengths_bp = _backward_tables(ranks_bev, ranks_depth, ranks_feat,
                                                             n_feat_rows)
        depth_grad = depth.new_zeros(depth.shape)
        feat_grad = feat.new_zeros(feat.shape)
        out_grad = out_grad.contiguous()
        bev_pool_v2_ext.bev_pool_v2_backward(out_grad, depth_grad, feat_grad, depth, feat, rd, rf,
                                             rb, lengths_bp, starts_bp)
        return depth_grad, feat_grad, None, None, None, None, None, None


def bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, interval_starts,
                interval_lengths):
    """Returns the pooled feature as a contiguous (B, C, Z, Y, X) tensor."""
    x = QuickCumsumCuda.apply(depth.float(), feat.float(), ranks_depth, ranks_feat, ranks_bev, bev_feat_shape,
                              interval_starts, interval_lengths)
    return x.permute(0, 4, 1, 2, 3).contiguous()


class TRTBEVPoolv2(torch.autograd.Function):
    """Export shim kept for API compatibility (reference :95-142): single-batch, Z collapsed."""

    @staticmethod
    def symbolic(g, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                 interval_lengths, out_height=128, out_width=128):
        return g.op("mmdeploy::bev_pool_v2", depth, feat, ranks_depth, ranks_feat, ranks_bev,
                    interval_starts, interval_lengths, out_height_i=out_height,
                    out_width_i=out_width)

    @staticmethod
    def forward(g, depth, feat, ranks_depth, ranks_feat, ranks_bev, interval_starts,
                interval_lengths, out_height=128, out_width=128):
        feat = feat.unsqueeze(0)      # (N,H,W,C)  -> (1,N,H,W,C)
        depth = depth.unsqueeze(0)    # (N,D,H,W)  -> (1,N,D,H,W)
        shape = (depth.shape[0], 1, out_height, out_width, feat.shape[-1])
        bev = bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev, shape, interval_starts,
                          interval_lengths)
        return bev.squeeze(2).permute(0, 2, 3, 1)
