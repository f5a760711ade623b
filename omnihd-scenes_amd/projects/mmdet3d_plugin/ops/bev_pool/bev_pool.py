"""v1 BEV pooling with the reference's Python API (ops/bev_pool/bev_pool.py:37-97).

The plugin imports this op at load time (projects/mmdet3d_plugin/__init__.py:20) although no
config calls it; it is provided so that the import succeeds and behaves.
"""
import torch

from omnihd_amd import ops as _ops

from . import bev_pool_ext

__all__ = ["bev_pool"]


class QuickCumsumCuda(torch.autograd.Function):
    """Segment-sum of rank-sorted point features into a (B,D,H,W,C) grid."""

    @staticmethod
    def forward(ctx, x, geom_feats, ranks, B, D, H, W):
        # intervals = runs of equal rank (reference :40-45), built on the device
        n = ranks.shape[0]
        key = ranks.contiguous().int()
        _, _, interval_starts, interval_lengths = _ops.sort_ranks(key, [], 32)
        geom_feats = geom_feats.contiguous().int()
        out = bev_pool_ext.bev_pool_forward(x.contiguous().float(), geom_feats, interval_lengths,
                                            interval_starts, B, D, H, W)
        ctx.save_for_backward(interval_starts, interval_lengths, geom_feats)
        ctx.saved_shapes = B, D, H, W
        return out

    @staticmethod
    def backward(ctx, out_grad):
        interval_starts, interval_lengths, geom_feats = ctx.saved_tensors
        B, D, H, W = ctx.saved_shapes
        x_grad = bev_pool_ext.bev_pool_backward(out_grad.contiguous(), geom_feats, interval_lengths,
                                                interval_starts, B, D, H, W)
        return x_grad, None, None, None, None, None, None


def bev_pool(feats, coords, B, D, H, W):
    """feats (N,C), coords (N,4) = (h_idx, w_idx, d_idx, b_idx) -> (B, C, D, H, W)."""
    assert feats.shape[0] == coords.shape[0]
    ranks = (coords[:, 0] * (W * D * B) + coords[:, 1] * (D * B) + coords[:, 2] * B + coords[:, 3])
    indices = ranks.argsort(stable=True)
    feats, coords, ranks = feats[indices], coords[indices], ranks[indices]
    x = QuickCumsumCuda.apply(feats, coords, ranks, B, D, H, W)
    return x.permute(0, 4, 1, 2, 3).contiguous()
