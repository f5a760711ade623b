"""GPU parity tests of the pooling plan that is built ON THE DEVICE without a host round trip (csrc/pool_plan.hip,
omnihd_amd/pool_plan.py) — SURVEY 8 a-3 as the reference runs it: new tables for a new calibration in every forward
(bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:283-300, :302-362; lidar2img per sample:
datasets/newscenes_dataset.py:203-216).

Bar: the rank tables are BIT-EXACT against the oracle (numpy restatement pinned to the reference's goldens) on jittered rigs,
against the golden vectors and the full-size checksums recorded from the reference; the forward / backward results on a
device-built plan are bit-identical to those on the host-built plan of the same calibration."""
import numpy as np
import pytest
import torch

from oracle import lss_oracle as O
from tests.helpers import PC_RANGE, full_size_geometry, t

pytestmark = pytest.mark.gpu


def _grid(pc, g):
    return O.gen_dx_bx([pc[0], pc[3], g], [pc[1], pc[4], g], [pc[2], pc[5], g])


def _to_yxz(rb, nx, B):
    X, Y, Z = (int(v) for v in nx)
    rb = rb.astype(np.int64)
    x, y, z, b = rb % X, (rb // X) % Y, (rb // (X * Y)) % Z, rb // (X * Y * Z)
    return ((b * Y + y) * X + x) * Z + z


def _assert_same_tables_as_host_plan(dp, hp):
    """Every table of the device-built plan ``dp`` against the host-built plan ``hp`` of the same geometry."""
    from omnihd_amd import plan as P
    c = dp.counts(wait=True)
    n = hp.n_points
    assert c["points"] == n and c["rows"] == hp.n_intervals and c["status"] == 0
    n_tiles = hp.tile_row.numel() - 1
    assert c["tiles"] == n_tiles and c["tiles_per_xcd"] == (n_tiles + 7) // 8
    assert torch.equal(dp.row_ptr, hp.row_ptr)
    pt, ivl_rel, desc32 = P.direct_tables(hp)
    assert torch.equal(dp.pt[:n], pt)
    assert torch.equal(dp.ivl_rel[:c["rows"]], ivl_rel)
    # the tile ORDER is a performance choice (azimuth in fp32 on the device, fp64 on the host): same descriptors as a set
    mine = dp.desc32[:8 * c["tiles_per_xcd"]].cpu().numpy()
    theirs = desc32.cpu().numpy()
    mine, theirs = mine[mine[:, 1] > 0], theirs[theirs[:, 1] > 0]
    assert mine.shape == theirs.shape == (n_tiles, 32)
    assert np.array_equal(mine[np.argsort(mine[:, 0])], theirs[np.argsort(theirs[:, 0])])
    assert torch.equal(dp.pix_ptr, hp.pix_ptr)
    assert torch.equal(dp.row_bin[:n], P._row_bin(hp))
    po = dp.patch_order.cpu().numpy()
    live = po[po >= 0]
    assert np.array_equal(np.sort(live), np.arange(dp.n_patch)), "every patch exactly once"
    runs = po.reshape(8, -1)
    assert all(np.all(r[:int((r >= 0).sum())] >= 0) for r in runs), "idle slots only behind a run's patches"
    assert c["patch_run"] == max(int((r >= 0).sum()) for r in runs) <= dp.patch_per


def _pool_both(dp, hp, depth, feat, keep=False):
    from omnihd_amd.plan import planned_pool
    res = []
    for plan in (dp, hp):
        d, f = depth.clone().requires_grad_(), feat.clone().requires_grad_()
        out = planned_pool(d, f, plan, keep_empty_rows=keep)
        w = torch.linspace(0.5, 1.5, out.numel(), device=out.device).view(out.shape[0], -1)
        (out.reshape(out.shape[0], -1) * w).sum().backward()
        res.append((out.detach().clone(), d.grad.clone(), f.grad.clone()))
    return res


@pytest.mark.parametrize("layout", ["byxz", "bzyx"])
def test_tiny_rig_device_plan_equals_host_plan_and_the_golden_tables(cuda, golden, layout):
    import omnihd_amd
    pc, g = golden["g2_pc_range"].tolist(), float(golden["g2_grid"])
    dx, bx, nx = _grid(pc, g)
    geom = t(golden["g2_geom"], cuda)
    B, N, D, H, W, _ = geom.shape
    dp = omnihd_amd.build_device_plan(dx, bx, nx, layout=layout, geom=geom, keep_sorted=True)
    hp = omnihd_amd.build_plan(geom, dx, bx, nx, layout=layout)
    _assert_same_tables_as_host_plan(dp, hp)
    rb, rd, rf = (x.cpu().numpy() for x in dp.reference_tables())
    if layout == "bzyx":                                   # the reference's numbering: the golden tables themselves
        assert np.array_equal(rb, golden["g3_ranks_bev"])
        assert np.array_equal(rd, golden["g3_ranks_depth"]) and np.array_equal(rf, golden["g3_ranks_feat"])
    else:
        assert np.array_equal(rb, np.sort(_to_yxz(golden["g3_ranks_bev"], nx, B)))
    rng = np.random.default_rng(5)
    depth = t(rng.random((B, N, D, H, W), dtype=np.float32), cuda)
    feat = t(rng.standard_normal((B, N, H, W, 64), dtype=np.float32), cuda)
    (o1, dg1, fg1), (o2, dg2, fg2) = _pool_both(dp, hp, depth, feat)
    assert torch.equal(o1, o2) and torch.equal(dg1, dg2) and torch.equal(fg1, fg2)


def test_adversarial_coordinates_bit_exact(cuda, golden):
    """Voxel edges, (-1,0) truncation (defect D3), NaN, +-1e30, out-of-range: the reference-format tables from a device plan."""
    import omnihd_amd
    dx, bx, nx = O.gen_dx_bx([-2.0, 2.0, 1.0], [-2.0, 2.0, 1.0], [-1.0, 1.0, 1.0])
    dp = omnihd_amd.build_device_plan(dx, bx, nx, layout="bzyx", geom=t(golden["g3adv_coor"], cuda), keep_sorted=True)
    rb, rd, rf = (x.cpu().numpy() for x in dp.reference_tables())
    assert np.array_equal(rb, golden["g3adv_ranks_bev"]) and np.array_equal(rd, golden["g3adv_ranks_depth"])
    assert np.array_equal(rf, golden["g3adv_ranks_feat"])
    assert dp.counts()["rows"] == golden["g3adv_starts"].size


def test_no_point_inside_the_grid_gives_zeros_without_asking_the_host(cuda):
    """Defect D4 (the reference crashes on an empty frustum): all-zero BEV, zero gradients — and the plan never needs its
    point count on the host."""
    import omnihd_amd
    from omnihd_amd.plan import planned_pool
    dx, bx, nx = O.gen_dx_bx([-2.0, 2.0, 1.0], [-2.0, 2.0, 1.0], [-1.0, 1.0, 1.0])
    coor = torch.full((1, 2, 3, 4, 16, 3), 100.0, device=cuda)
    dp = omnihd_amd.build_device_plan(dx, bx, nx, geom=coor)
    depth = torch.rand(1, 2, 3, 4, 16, device=cuda, requires_grad=True)
    feat = torch.randn(1, 2, 4, 16, 64, device=cuda, requires_grad=True)
    out = planned_pool(depth, feat, dp)
    out.sum().backward()
    assert float(out.abs().sum()) == 0.0 and float(depth.grad.abs().sum()) == 0.0 and float(feat.grad.abs().sum()) == 0.0
    assert dp.counts(wait=True)["points"] == 0


def _jittered_inverse(l2i, rng, yaw_deg=1.0, shift=0.5):
    """lidar2img of a frame whose ego pose differs a little (the reference composes it through the ego poses of the camera and
    the LiDAR sweep: newscenes_devkit/newscenes_converter_final.py:346-383) -> fp32 rots / trans as the detector computes them
    (torch.Tensor(mat).inverse(), bevf_faster_rcnn_bevdepth.py:121-130)."""
    a = np.radians(rng.uniform(-yaw_deg, yaw_deg))
    T = np.eye(4)
    T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    T[:3, 3] = rng.uniform(-shift, shift, 3) * [1, 1, 0.1]
    inv = torch.Tensor(np.stack([m @ T for m in l2i])).inverse()
    return inv[:, :3, :3][None].contiguous(), inv[:, :3, 3][None].contiguous()


@pytest.mark.parametrize("H,W,fx,n_rigs", [(128, 352, 205.0, 10), (256, 704, 410.0, 2)])
def test_jittered_rigs_fused_geometry_tables_bit_exact_against_the_oracle(cuda, H, W, fx, n_rigs):
    """A new calibration per frame: the plan is built from rots / trans (the frustum point is formed inside the key kernel with
    the rounding steps of get_geometry, :235-264) and its tables equal the oracle's voxel_pooling_prepare_v2 on the oracle's
    geometry, bit for bit — ten rigs at half size, two at R1."""
    import omnihd_amd
    dx, bx, nx = _grid(PC_RANGE, 0.5)
    fr = O.create_frustum((H, W), 4, [1, 60, 1])
    xs, ys, ds = O.frustum_axes((H, W), 4, [1, 60, 1])
    axes = tuple(t(np.asarray(a, dtype=np.float32), cuda) for a in (xs, ys, ds))
    rng = np.random.default_rng(2026)
    l2i = O.synthetic_rig(H, W, fx)
    for _ in range(n_rigs):
        rots, trans = _jittered_inverse(l2i, rng)
        geom = O.get_geometry(fr, rots.numpy(), trans.numpy())
        want = O.voxel_pooling_prepare_v2(geom, dx, bx, nx)
        dp = omnihd_amd.build_device_plan(dx, bx, nx, layout="bzyx", rots=rots.to(cuda), trans=trans.to(cuda), axes=axes,
                                          keep_sorted=True)
        rb, rd, rf = (x.cpu().numpy() for x in dp.reference_tables())
        assert np.array_equal(rb, want[0]) and np.array_equal(rd, want[1]) and np.array_equal(rf, want[2])
        assert dp.counts()["rows"] == want[3].size
        # the backward tables are the stable re-sort by pixel (ops/bev_pool_v2/bev_pool.py:47-57)
        bp = O.backward_tables(want[0], want[1], want[2])
        fhw = (H // 4) * (W // 4)
        assert np.array_equal(dp.row_bin[:rb.size].cpu().numpy(), bp[0] | (((bp[1] // fhw) % 59) << 24))
        assert np.array_equal(np.diff(dp.pix_ptr.cpu().numpy()), np.bincount(want[2], minlength=6 * fhw))


@pytest.mark.parametrize("tag,H,W", [("r1", 256, 704), ("r2", 544, 960)])
def test_full_size_module_path_reproduces_the_reference_checksums_and_the_host_plan(cuda, golden, tag, H, W):
    """The plan the LSS module builds through its own cache path for the 6-camera R1 / R2 rigs carries the counts and table
    checksums recorded from the reference (:302-362, tests/golden/make_golden.py) and equals the host-built plan table by
    table; pooled features and both gradients are bit-identical on the two plans."""
    import omnihd_amd
    from omnihd_amd import pool_plan
    from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet import LiftSplatShoot_Depth
    net = LiftSplatShoot_Depth(final_dim=(H, W), camera_depth_range=[1, 60, 1], pc_range=PC_RANGE, downsample=4, grid=0.5,
                               inputC=256, camC=64, norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01))
    net.frustum.data = net.frustum.data.to(cuda)
    rots, trans = t(golden[f"full_{tag}_rots"], cuda), t(golden[f"full_{tag}_trans"], cuda)
    dp = net._plan_for(rots, trans, (None, None, None, None))
    assert isinstance(dp, pool_plan.DevicePoolPlan) and dp.layout == "byxz"
    assert net._plan_for(rots, trans, (None, None, None, None)) is dp
    cs_ref = golden[f"full_{tag}_checksums"].tolist()
    c = dp.counts(wait=True)
    assert c["points"] == cs_ref[0] and c["rows"] == cs_ref[1]
    geom, dx, bx, nx = full_size_geometry(tag)
    sp = omnihd_amd.build_device_plan(dx, bx, nx, layout="bzyx", rots=rots, trans=trans, axes=net._frustum_axes(cuda),
                                      keep_sorted=True)
    rb, rd, rf = sp.reference_tables()
    assert [int(rb.long().sum()), int(rd.long().sum()), int(rf.long().sum())] == cs_ref[2:5]
    hp = omnihd_amd.build_plan(t(geom, cuda), dx, bx, nx, layout="byxz")
    _assert_same_tables_as_host_plan(dp, hp)
    if tag == "r1":
        rng = np.random.default_rng(11)
        depth = t(rng.random((1, 6, 59, H // 4, W // 4), dtype=np.float32), cuda)
        feat = t(rng.standard_normal((1, 6, H // 4, W // 4, 64), dtype=np.float32), cuda)
        (o1, dg1, fg1), (o2, dg2, fg2) = _pool_both(dp, hp, depth, feat)
        assert torch.equal(o1, o2) and torch.equal(dg1, dg2) and torch.equal(fg1, fg2)


def test_kept_output_buffer_across_calibrations_zeroes_exactly_the_rows_that_emptied(cuda):
    """Per-frame calibrations with keep_empty_rows: the buffer filled under calibration A is handed to calibration B, whose
    launch zero-fills only the rows that A occupied and B does not.  Results equal fresh buffers, also when going back to A."""
    import omnihd_amd
    from omnihd_amd import plan as P, pool_plan
    from omnihd_amd.plan import planned_pool
    H, W, fx = 64, 176, 102.5
    dx, bx, nx = _grid(PC_RANGE, 0.5)
    xs, ys, ds = O.frustum_axes((H, W), 4, [1, 60, 1])
    axes = tuple(t(np.asarray(a, dtype=np.float32), cuda) for a in (xs, ys, ds))
    rng = np.random.default_rng(7)
    l2i = O.synthetic_rig(H, W, fx)
    plans = []
    for _ in range(3):
        rots, trans = _jittered_inverse(l2i, rng, yaw_deg=3.0, shift=1.5)
        plans.append(omnihd_amd.build_device_plan(dx, bx, nx, rots=rots.to(cuda), trans=trans.to(cuda), axes=axes))
    occ = [np.diff(p.row_ptr.cpu().numpy()) > 0 for p in plans]
    assert (occ[0] & ~occ[1]).sum() > 0 and (occ[1] & ~occ[0]).sum() > 0, "the jitter must move rows in and out of the frustum"
    depth = t(rng.random((1, 6, 59, H // 4, W // 4), dtype=np.float32), cuda)
    feat = t(rng.standard_normal((1, 6, H // 4, W // 4, 64), dtype=np.float32), cuda)
    fresh = [planned_pool(depth, feat, p).clone() for p in plans]
    pool_plan._FAMILY.clear()
    before = P.FAST_PATHS["kept_output"]
    for k in (0, 1, 2, 0, 0, 1):
        got = planned_pool(depth, feat, plans[k], keep_empty_rows=True)
        assert torch.equal(got, fresh[k]), k
        del got
    assert P.FAST_PATHS["kept_output"] - before == 6
    fam = pool_plan._FAMILY[(cuda.index or 0, plans[0].n_rows, 64)]
    assert len(fam) == 1, "one buffer serves every calibration while nobody holds a result"


def test_build_and_use_enqueue_without_any_synchronisation(cuda):
    """The whole cache-miss path — build, forward, backward — under torch's synchronisation detector (the library call itself has
    no synchronising HIP call by construction: tests/test_abi.py greps its source)."""
    import omnihd_amd
    from omnihd_amd.plan import planned_pool
    H, W, fx = 64, 176, 102.5
    dx, bx, nx = _grid(PC_RANGE, 0.5)
    xs, ys, ds = O.frustum_axes((H, W), 4, [1, 60, 1])
    axes = tuple(t(np.asarray(a, dtype=np.float32), cuda) for a in (xs, ys, ds))
    rots, trans = _jittered_inverse(O.synthetic_rig(H, W, fx), np.random.default_rng(3))
    rots, trans = rots.to(cuda), trans.to(cuda)
    depth = torch.rand(1, 6, 59, H // 4, W // 4, device=cuda, requires_grad=True)
    feat = torch.randn(1, 6, H // 4, W // 4, 64, device=cuda, requires_grad=True)
    omnihd_amd.build_device_plan(dx, bx, nx, rots=rots, trans=trans, axes=axes)           # sizes / walks cached, pinned pool warm
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        dp = omnihd_amd.build_device_plan(dx, bx, nx, rots=rots, trans=trans, axes=axes)
        out = planned_pool(depth, feat, dp, keep_empty_rows=True)
        out.sum().backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert float(out.detach().abs().sum()) > 0
