"""GPU parity tests of the rank-table preparation (voxel_pooling_prepare_v2 on the device):
bit-exact tables vs the golden vectors captured from the reference and vs the numpy oracle, and
full-size (R1/R2) checksums recorded from the reference."""
import numpy as np
import pytest
import torch

from oracle import lss_oracle as O
from tests.helpers import full_size_geometry, t

pytestmark = pytest.mark.gpu
NAMES = ["ranks_bev", "ranks_depth", "ranks_feat", "starts", "lengths"]


def test_tiny_rig_tables_bit_exact(cuda, golden):
    from omnihd_amd import ops
    pc, g = golden["g2_pc_range"].tolist(), float(golden["g2_grid"])
    dx, bx, nx = O.gen_dx_bx([pc[0], pc[3], g], [pc[1], pc[4], g], [pc[2], pc[5], g])
    tabs = ops.voxel_pooling_prepare_v2(t(golden["g2_geom"], cuda), dx, bx, nx)
    for k, tb in zip(NAMES, tabs):
        assert tb.dtype == torch.int32 and tb.is_contiguous()
        assert np.array_equal(tb.cpu().numpy(), golden[f"g3_{k}"]), k


def test_adversarial_coordinates_bit_exact(cuda, golden):
    """Voxel edges, (-1,0) truncation (defect D3), NaN, +-1e30, out-of-range."""
    from omnihd_amd import ops
    dx, bx, nx = O.gen_dx_bx([-2.0, 2.0, 1.0], [-2.0, 2.0, 1.0], [-1.0, 1.0, 1.0])
    tabs = ops.voxel_pooling_prepare_v2(t(golden["g3adv_coor"], cuda), dx, bx, nx)
    for k, tb in zip(NAMES, tabs):
        assert np.array_equal(tb.cpu().numpy(), golden[f"g3adv_{k}"]), k


def test_no_point_in_grid_returns_none(cuda):
    from omnihd_amd import ops
    dx, bx, nx = O.gen_dx_bx([-2.0, 2.0, 1.0], [-2.0, 2.0, 1.0], [-1.0, 1.0, 1.0])
    coor = torch.full((1, 1, 2, 2, 2, 3), 100.0, device=cuda)
    assert ops.voxel_pooling_prepare_v2(coor, dx, bx, nx) == (None,) * 5


@pytest.mark.parametrize("grid", [0.6, 0.7])
def test_inexact_division_grid_matches_oracle(cuda, grid):
    """dx that is not a power of two: true fp32 division, no reciprocal."""
    from omnihd_amd import ops
    rng = np.random.default_rng(17)
    pc = [-6.0, -4.2, -1.4, 6.0, 4.2, 1.4]
    dx, bx, nx = O.gen_dx_bx([pc[0], pc[3], grid], [pc[1], pc[4], grid], [pc[2], pc[5], grid])
    coor = (rng.random((2, 2, 5, 6, 7, 3), dtype=np.float32) * 16 - 8).astype(np.float32)
    # put many points exactly on voxel edges
    edges = (bx[0] - dx[0] / np.float32(2)) + dx[0] * rng.integers(0, nx[0], size=200).astype(np.float32)
    coor.reshape(-1, 3)[:200, 0] = edges
    want = O.voxel_pooling_prepare_v2(coor, dx, bx, nx)
    got = ops.voxel_pooling_prepare_v2(t(coor, cuda), dx, bx, nx)
    for k, g, w in zip(NAMES, got, want):
        assert np.array_equal(g.cpu().numpy(), w), k


@pytest.mark.parametrize("tag", ["r1", "r2"])
def test_full_size_checksums_recorded_from_reference(cuda, golden, tag):
    from omnihd_amd import ops
    geom, dx, bx, nx = full_size_geometry(tag)
    tabs = ops.voxel_pooling_prepare_v2(t(geom, cuda), dx, bx, nx)
    cs = [tabs[0].numel(), tabs[3].numel()] + [int(x.long().sum()) for x in tabs] + [int(tabs[4].max())]
    assert cs == golden[f"full_{tag}_checksums"].tolist()
    rb, rd, st, ln = (x.cpu().numpy() for x in (tabs[0], tabs[1], tabs[3], tabs[4]))
    # size-independent properties: sorted keys, canonical order, intervals tile the point list
    assert np.all(np.diff(rb) >= 0)
    head = np.zeros(rb.size, bool); head[st] = True
    assert np.all((np.diff(rd) > 0) | head[1:])
    assert st[0] == 0 and np.array_equal(st[1:], np.cumsum(ln)[:-1]) and ln.sum() == rb.size
    assert np.all(rb[st][1:] > rb[st][:-1])


def test_plan_layouts_agree(cuda, golden):
    """build_plan (fused keys -> CSR) for both row numberings vs the reference-format tables."""
    import omnihd_amd
    pc, g = golden["g2_pc_range"].tolist(), float(golden["g2_grid"])
    dx, bx, nx = O.gen_dx_bx([pc[0], pc[3], g], [pc[1], pc[4], g], [pc[2], pc[5], g])
    geom = t(golden["g2_geom"], cuda)
    B = geom.shape[0]
    X, Y, Z = (int(v) for v in nx)
    p = omnihd_amd.build_plan(geom, dx, bx, nx, layout="bzyx")
    assert np.array_equal(p.ranks_row.cpu().numpy(), golden["g3_ranks_bev"])
    assert np.array_equal(p.ranks_depth.cpu().numpy(), golden["g3_ranks_depth"])
    assert np.array_equal(p.ranks_feat.cpu().numpy(), golden["g3_ranks_feat"])
    rp = p.row_ptr.cpu().numpy()
    assert rp[0] == 0 and rp[-1] == p.n_points and np.all(np.diff(rp) >= 0)
    counts = np.bincount(golden["g3_ranks_bev"], minlength=B * Z * Y * X)
    assert np.array_equal(np.diff(rp), counts)
    q = omnihd_amd.build_plan(geom, dx, bx, nx, layout="byxz")
    rb = golden["g3_ranks_bev"].astype(np.int64)
    x, y, z, b = rb % X, (rb // X) % Y, (rb // (X * Y)) % Z, rb // (X * Y * Z)
    perm = ((b * Y + y) * X + x) * Z + z
    assert np.array_equal(np.sort(perm), q.ranks_row.cpu().numpy())
