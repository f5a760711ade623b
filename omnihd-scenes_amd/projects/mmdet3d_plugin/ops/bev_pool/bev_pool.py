"""v1 BEV pooling under the reference's Python name (ops/bev_pool/bev_pool.py:86-97: ``bev_pool(feats, coords, B, D, H, W)``).

The plugin imports this op at load time (projects/mmdet3d_plugin/__init__.py:20) although no config calls it; it is
provided so that the import succeeds and behaves.  Points are ordered by their cell with a stable device sort, the
runs of equal cells become the kernel's intervals (rocPRIM scan, ``omnihd_amd.ops.sort_ranks``), and the segment sums
run in csrc/bev_pool_v1.hip.
"""
import torch

from omnihd_amd import ops as _ops

from . import bev_pool_ext

__all__ = ["bev_pool"]


def _cell_rank(coords, B, D, H, W):
    """Linear cell index in the reference's (h, w, d, b) significance order (:89-94)."""
    h, w, d, b = coords.unbind(dim=1)
    return ((h * W + w) * D + d) * B + b


class _SegmentPool(torch.autograd.Function):
    """(N, C) features sorted by cell -> (B, D, H, W, C) grid of per-cell sums; gradient = gather."""

    @staticmethod
    def forward(ctx, feats, cells, ranks, grid):
        starts, lengths = _ops.sort_ranks(ranks.contiguous().int(), [], 32)[2:]
        cells = cells.contiguous().int()
        ctx.save_for_backward(starts, lengths, cells)
        ctx.grid = grid
        return bev_pool_ext.bev_pool_forward(feats.contiguous().float(), cells, lengths, starts, *grid)

    @staticmethod
    def backward(ctx, grad):
        starts, lengths, cells = ctx.saved_tensors
        return bev_pool_ext.bev_pool_backward(grad.contiguous(), cells, lengths, starts, *ctx.grid), None, None, None


# the reference's class name for this function object (ops/bev_pool/bev_pool.py:37)
QuickCumsumCuda = _SegmentPool


def bev_pool(feats, coords, B, D, H, W):
    """feats (N, C), coords (N, 4) = (h_idx, w_idx, d_idx, b_idx) -> (B, C, D, H, W)."""
    if feats.shape[0] != coords.shape[0]:
        raise AssertionError("feats and coords disagree on the number of points")
    ranks = _cell_rank(coords, B, D, H, W)
    order = ranks.argsort(stable=True)
    pooled = _SegmentPool.apply(feats[order], coords[order], ranks[order], (B, D, H, W))
    return pooled.permute(0, 4, 1, 2, 3).contiguous()
