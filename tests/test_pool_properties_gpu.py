"""Size-independent properties of bev_pool_v2 at the full frame sizes R1 (256x704) and R2 (544x960), through the path the model
runs (``planned_pool``: k_pool_fwd_direct, k_pool_bwd_patch).  They need no oracle run at
2-4.5 M points and hold for ANY correct implementation of the reference's operator (ops/bev_pool_v2/bev_pool.py:9-57):
  * checksum:  sum over rows of out = sum over points of depth * feat   (per channel; an independent torch gather-sum);
  * adjoint:   <out, og> = <depth, depth_grad> = <feat, feat_grad>        (out is bilinear in (depth, feat));
  * linearity: pool(depth, a*f1 + f2) = a*pool(depth, f1) + pool(depth, f2), the same in depth;
  * the plan:  rows sorted, every frustum point that the reference keeps appears exactly once, rebuilt tables identical."""
import numpy as np
import pytest
import torch

from tests.helpers import full_size_geometry, t

FULL = {"r1": (64, 176, 2025022), "r2": (136, 240, 4503872)}      # fH, fW, points the reference keeps
pytestmark = pytest.mark.gpu


def _setup(cuda, res, seed):
    from omnihd_amd import build_plan
    fH, fW, n_ref = FULL[res]
    geom, dx, bx, nx = full_size_geometry(res)
    plan = build_plan(t(geom, cuda), dx, bx, nx, layout="byxz")
    assert plan.n_points == n_ref
    g = torch.Generator(device="cpu").manual_seed(seed)
    depth = torch.rand(1, 6, 59, fH, fW, generator=g).softmax(2).to(cuda)
    feat = torch.randn(1, 6, fH, fW, 64, generator=g).to(cuda)
    og = torch.randn(plan.n_rows, 64, generator=g).to(cuda)
    return plan, depth, feat, og


def _dot(a, b):
    return float((a.double().reshape(-1) * b.double().reshape(-1)).sum())


@pytest.mark.parametrize("res", ["r1", "r2"])
def test_checksum_of_the_pooled_rows(cuda, res):
    from omnihd_amd.plan import planned_pool
    plan, depth, feat, og = _setup(cuda, res, 21)
    out = planned_pool(depth, feat, plan)                              # (B, C, Z, Y, X) view
    got = out.double().sum(dim=(0, 2, 3, 4))
    want = torch.zeros(64, dtype=torch.float64, device=cuda)
    rd, rf = plan.ranks_depth.long(), plan.ranks_feat.long()
    for a in range(0, rd.numel(), 1 << 19):                            # chunks: no 1 GB temporary
        d = depth.reshape(-1)[rd[a:a + (1 << 19)]].double()
        want += (d[:, None] * feat.reshape(-1, 64)[rf[a:a + (1 << 19)]].double()).sum(0)
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max() + 1.0)
    # the operator writes exactly the rows the tables name
    touched = torch.zeros(plan.n_rows, dtype=torch.bool, device=cuda)
    touched[plan.ranks_row.long()] = True
    rows = out.permute(0, 3, 4, 2, 1).reshape(plan.n_rows, 64)       # 'byxz' memory order = row order
    assert not bool(rows[~touched].any())


@pytest.mark.parametrize("res", ["r1", "r2"])
def test_forward_and_backward_are_adjoint(cuda, res):
    from omnihd_amd.plan import planned_pool
    plan, depth, feat, og = _setup(cuda, res, 22)
    depth.requires_grad_(); feat.requires_grad_()
    out = planned_pool(depth, feat, plan)
    rows = out.permute(0, 3, 4, 2, 1).reshape(plan.n_rows, 64)
    rows.backward(og)
    lhs = _dot(rows.detach(), og)
    assert abs(_dot(depth.detach(), depth.grad) - lhs) <= 2e-6 * abs(lhs) + 1e-3
    assert abs(_dot(feat.detach(), feat.grad) - lhs) <= 2e-6 * abs(lhs) + 1e-3


@pytest.mark.parametrize("res", ["r1", "r2"])
def test_forward_is_linear_in_each_argument(cuda, res):
    from omnihd_amd.plan import planned_pool
    plan, depth, f1, og = _setup(cuda, res, 23)
    f2 = torch.randn_like(f1)
    d2 = torch.rand_like(depth)
    a = 0.37
    p = lambda d, f: planned_pool(d, f, plan).clone()
    base = p(depth, f1)
    for lhs, rhs in ((p(depth, a * f1 + f2), a * base + p(depth, f2)), (p(a * depth + d2, f1), a * base + p(d2, f1))):
        assert float((lhs - rhs).abs().max()) <= 4e-6 * float(lhs.abs().max())          # fp32 rounding of sums of <= a few hundred terms
    assert not bool(p(torch.zeros_like(depth), f1).any())             # and maps zero to zero exactly


@pytest.mark.parametrize("res", ["r1", "r2"])
def test_plan_tables_are_a_sorted_permutation_and_rebuild_identically(cuda, res):
    from omnihd_amd import build_plan
    fH, fW, n_ref = FULL[res]
    geom, dx, bx, nx = full_size_geometry(res)
    p1 = build_plan(t(geom, cuda), dx, bx, nx, layout="byxz")
    p2 = build_plan(t(geom, cuda), dx, bx, nx, layout="byxz")
    rr = p1.ranks_row.long()
    assert bool((rr[1:] >= rr[:-1]).all())                                               # sorted by output row
    assert int(torch.unique(p1.ranks_depth).numel()) == p1.n_points == n_ref             # every kept frustum point exactly once
    assert bool((p1.ranks_feat.long() == (p1.ranks_depth.long() // (59 * fH * fW)) * fH * fW + p1.ranks_depth.long() % (fH * fW)).all())
    for name in ("ranks_row", "ranks_depth", "ranks_feat", "row_ptr", "bp_ranks_row", "bp_ranks_depth", "pix_ptr"):
        assert torch.equal(getattr(p1, name), getattr(p2, name)), name                   # idempotent: same geometry, same tables
    # backward tables: the same points, grouped by pixel, inside a pixel by row
    assert torch.equal(torch.sort(p1.bp_ranks_depth)[0], torch.sort(p1.ranks_depth)[0])
    counts = (p1.pix_ptr[1:] - p1.pix_ptr[:-1]).long()
    assert int(counts.sum()) == n_ref and int(counts.max()) <= 59
