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


@pytest.mark.parametrize("tag,H,W", [("r1", 256, 704), ("r2", 544, 960)])
def test_device_geometry_of_the_lss_module_is_bit_exact_and_its_plan_reproduces_the_reference_checksums(cuda, golden, tag, H, W):
    """the tables north_star wants bit-exact are built from the geometry the PRODUCT computes on the
    device (LiftSplatShoot_Depth.get_geometry, three broadcast multiply-adds per axis), not from a host geometry.  For the
    6-camera R1 and R2 rigs: the device geometry equals the oracle's (= the reference's torch-CPU get_geometry,
    cam_stream_lss_bevpoolv2_depthnet.py:235-264) bit for bit, the plan the module builds from it through its own cache
    path (`_plan_for`) carries the reference's point / interval counts and table checksums (:302-362, recorded from the
    reference in tests/golden/make_golden.py), and the reference-format tables from the same device geometry do too."""
    from omnihd_amd import ops
    from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet import LiftSplatShoot_Depth
    from tests.helpers import PC_RANGE
    net = LiftSplatShoot_Depth(final_dim=(H, W), camera_depth_range=[1, 60, 1], pc_range=PC_RANGE, downsample=4, grid=0.5,
                               inputC=256, camC=64, norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01))
    net.frustum.data = net.frustum.data.to(cuda)               # only the geometry path is exercised: no need to move the convs
    rots, trans = t(golden[f"full_{tag}_rots"], cuda), t(golden[f"full_{tag}_trans"], cuda)
    with torch.no_grad():
        geom = net.get_geometry(rots, trans)
    want, dx, bx, nx = full_size_geometry(tag)
    assert geom.dtype == torch.float32 and tuple(geom.shape) == want.shape
    assert torch.equal(geom.cpu(), torch.from_numpy(want)), "device geometry differs from the reference's torch-CPU geometry"
    cs_ref = golden[f"full_{tag}_checksums"].tolist()
    tabs = ops.voxel_pooling_prepare_v2(geom.contiguous(), dx, bx, nx)
    cs = [tabs[0].numel(), tabs[3].numel()] + [int(x.long().sum()) for x in tabs] + [int(tabs[4].max())]
    assert cs == cs_ref
    # the host-scheduled plan of the same geometry (the module's own cache path builds its plans on the device since round 6:
    # tests/test_device_plan_gpu.py holds that path to the same checksums and to this plan, table by table)
    import omnihd_amd
    plan = omnihd_amd.build_plan(geom.contiguous(), dx, bx, nx, layout="byxz")
    assert plan.layout == "byxz" and plan.n_points == cs_ref[0] and plan.n_intervals == cs_ref[1]
    assert int(plan.ranks_depth.long().sum()) == cs_ref[3] and int(plan.ranks_feat.long().sum()) == cs_ref[4]
    assert int(plan.interval_lengths.long().sum()) == cs_ref[6] and int(plan.interval_lengths.max()) == cs_ref[7]
    # the plan numbers rows (b,y,x,z); mapped back to the reference's (b,z,y,x) numbering the row sum is the reference's too
    X, Y, Z = (int(v) for v in nx)
    r = plan.ranks_row.long()
    z, x, y, b = r % Z, (r // Z) % X, (r // (Z * X)) % Y, r // (Z * X * Y)
    assert int((((b * Z + z) * Y + y) * X + x).sum()) == cs_ref[2]
