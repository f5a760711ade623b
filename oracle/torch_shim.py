"""oracle/torch_shim.py — TEST INFRASTRUCTURE, NOT PRODUCT.

Lets the detector's Python host code run on CPU tensors by temporarily routing the four HIP-backed
operators to the CPU oracle (numpy rank tables, C pooling kernels, sequential voxelise, index
scatter, sequential rotated NMS).  Used ONLY by tests (CPU-side checks of the host logic, world_size-2 gloo runs) and by
bench.py's cpu_baseline leg; the product never imports it and has no CPU path of its own.
"""
import contextlib

import torch

from . import cpu as OC
from . import lss_oracle as O


class CpuPlan:
    """Rank tables in the reference's format, rebuilt on every call like the reference does."""

    def __init__(self, coor, dx, bx, nx):
        self.tabs = O.voxel_pooling_prepare_v2(coor.detach().cpu().numpy(), dx, bx, nx)
        self.n_points = 0 if self.tabs[0] is None else len(self.tabs[0])
        self.nx = [int(v) for v in nx]


class _CpuPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, feat, plan):
        rb, rd, rf, st, ln = plan.tabs
        B, C = depth.shape[0], feat.shape[-1]
        X, Y, Z = plan.nx
        out = OC.bev_pool_v2_fwd(depth.numpy(), feat.numpy(), rd, rf, rb, (B, Z, Y, X, C), st, ln, threads=True)
        ctx.save_for_backward(depth, feat)
        ctx.plan = plan
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, g):
        depth, feat = ctx.saved_tensors
        rb, rd, rf, st, ln = ctx.plan.tabs
        bp = O.backward_tables(rb, rd, rf)          # re-sort per backward (ops/bev_pool_v2/bev_pool.py:47-57)
        dg, fg = OC.bev_pool_v2_bwd(g.contiguous().numpy(), depth.numpy(), feat.numpy(), bp[1], bp[2], bp[0], bp[3],
                                    bp[4], threads=True)
        return torch.from_numpy(dg), torch.from_numpy(fg), None


def planned_pool(depth, feat, plan, keep_empty_rows=False):
    out = _CpuPool.apply(depth.float().contiguous(), feat.float().contiguous(), plan)
    return out.permute(0, 4, 1, 2, 3).contiguous()               # ops/bev_pool_v2/bev_pool.py:91


def hard_voxelize(points, voxel_size, pcr, max_points, max_voxels):
    v, c, n = OC.hard_voxelize(points.detach().numpy(), voxel_size, pcr, max_points, max_voxels)
    return torch.from_numpy(v), torch.from_numpy(c), torch.from_numpy(n)


def pillar_scatter(feats, coors, batch, ny, nx, channels_last=False):
    canvas = feats.new_zeros(batch, feats.shape[1], ny * nx)
    c = coors.long()
    for b in range(batch):                                       # PointPillarsScatter.forward_batch
        m = c[:, 0] == b
        canvas[b][:, c[m, 2] * nx + c[m, 3]] = feats[m].t()
    return canvas.view(batch, -1, ny, nx)


def nms_rotated(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    order = scores.sort(0, descending=True)[1]                   # torch's order, as the product wrapper uses
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep = OC.nms_rotated_sorted(boxes[order].detach().numpy(), float(thresh))
    kept = order[torch.from_numpy(keep)]
    return kept if post_max_size is None else kept[:post_max_size]


@contextlib.contextmanager
def oracle_ops():
    """Inside the block ``omnihd_amd`` ops used by the detector run on the CPU oracle."""
    import omnihd_amd
    from omnihd_amd import ops as gops
    from projects.mmdet3d_plugin.bevfusion.detectors import cam_stream_lss_bevpoolv2_depthnet as lssmod
    saved = (omnihd_amd.build_plan, lssmod.planned_pool, gops.hard_voxelize, gops.pillar_scatter, gops.nms_rotated)
    omnihd_amd.build_plan = lambda coor, dx, bx, nx, layout="bzyx", **_kw: CpuPlan(coor, dx, bx, nx)
    lssmod.planned_pool = planned_pool
    gops.hard_voxelize, gops.pillar_scatter, gops.nms_rotated = hard_voxelize, pillar_scatter, nms_rotated
    try:
        yield
    finally:
        (omnihd_amd.build_plan, lssmod.planned_pool, gops.hard_voxelize, gops.pillar_scatter,
         gops.nms_rotated) = saved
