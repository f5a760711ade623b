"""GPU parity tests of bev_pool_v2 / bev_pool (v1): HIP kernels through the C ABI vs the CPU
oracle, the reference's known-answer test, the golden vectors, and — when oracle/_ref was built —
the reference's own kernels compiled by hipcc.

Tolerances: index tables bit-exact; pooled features: bit-exact when the summation order is the
reference's (intervals <= 512 points), otherwise <= 1e-5 relative (north_star allows 1e-3)."""
import numpy as np
import pytest
import torch

from oracle import cpu as OC
from oracle import lss_oracle as O
from tests.helpers import RefKernels, full_size_geometry, random_tables, t

pytestmark = pytest.mark.gpu


def run_fwd(dev, depth, feat, rd, rf, rb, shape, st, ln):
    from projects.mmdet3d_plugin.ops.bev_pool_v2 import bev_pool_v2_ext as ext
    out = torch.zeros(shape, dtype=torch.float32, device=dev)
    ext.bev_pool_v2_forward(t(depth, dev), t(feat, dev), out, t(rd, dev), t(rf, dev), t(rb, dev), t(ln, dev), t(st, dev))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_reference_known_answer(cuda):
    """The reference's test_bev_pool_v2 (ops/bev_pool_v2/bev_pool.py:145-176), same values."""
    from projects.mmdet3d_plugin.ops.bev_pool_v2.bev_pool import bev_pool_v2
    depth = torch.tensor([0.3, 0.4, 0.2, 0.1, 0.7, 0.6, 0.8, 0.9], device=cuda).view(1, 1, 2, 2, 2).requires_grad_()
    feat = torch.ones(1, 1, 2, 2, 2, device=cuda).requires_grad_()
    rd = torch.tensor([0, 4, 1, 6], dtype=torch.int32, device=cuda)
    rf = torch.tensor([0, 0, 1, 2], dtype=torch.int32, device=cuda)
    rb = torch.tensor([0, 0, 1, 1], dtype=torch.int32, device=cuda)
    kept = torch.ones(4, dtype=torch.bool, device=cuda)
    kept[1:] = rb[1:] != rb[:-1]
    st = torch.where(kept)[0].int()
    ln = torch.zeros_like(st)
    ln[:-1] = st[1:] - st[:-1]
    ln[-1] = 4 - st[-1]
    bev = bev_pool_v2(depth, feat, rd, rf, rb, (1, 1, 2, 2, 2), st, ln)
    loss = bev.sum()
    loss.backward()
    assert loss == 4.4
    assert depth.grad.allclose(torch.tensor([2., 2., 0., 0., 2., 0., 2., 0.], device=cuda).view(1, 1, 2, 2, 2))
    assert feat.grad.allclose(torch.tensor([1.0, 1.0, 0.4, 0.4, 0.8, 0.8, 0., 0.], device=cuda).view(1, 1, 2, 2, 2))


def test_golden_forward_backward_through_plugin_api(cuda, golden):
    """Tiny rig: same inputs as the reference autograd Function saw; canonical tables."""
    from projects.mmdet3d_plugin.ops.bev_pool_v2.bev_pool import bev_pool_v2
    depth = t(golden["g4_depth"], cuda).requires_grad_()
    feat = t(golden["g4_feat"], cuda).requires_grad_()
    B, C, Z, Y, X = golden["g4_bev"].shape
    args = [t(golden[f"g3_{k}"], cuda) for k in ("ranks_depth", "ranks_feat", "ranks_bev")]
    nx = torch.tensor([X, Y, Z])          # 0-d tensors in the shape, as the reference passes them
    bev = bev_pool_v2(depth, feat, *args, (B, nx[2], nx[1], nx[0], C), t(golden["g3_starts"], cuda), t(golden["g3_lengths"], cuda))
    assert bev.shape == (B, C, Z, Y, X) and bev.is_contiguous()
    (bev * t(golden["g4_w"], cuda)).sum().backward()
    np.testing.assert_allclose(bev.detach().cpu().numpy(), golden["g4_bev"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(depth.grad.cpu().numpy(), golden["g4_depth_grad"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(feat.grad.cpu().numpy(), golden["g4_feat_grad"], rtol=1e-5, atol=1e-5)


def test_golden_raw_order_is_bit_exact(cuda, golden):
    """With the tables in the exact order the reference produced, the fma chain is identical."""
    B, C, Z, Y, X = golden["g4_bev"].shape
    out = run_fwd(cuda, golden["g4_depth"], golden["g4_feat"], golden["g3raw_ranks_depth"], golden["g3raw_ranks_feat"],
                  golden["g3raw_ranks_bev"], (B, Z, Y, X, C), golden["g3_starts"], golden["g3_lengths"])
    assert np.array_equal(out.transpose(0, 4, 1, 2, 3), golden["g4_bev"])


@pytest.mark.parametrize("c", [1, 2, 4, 8, 16, 32, 64, 80, 128, 256])
def test_forward_matches_oracle_all_channel_counts(cuda, c):
    rng = np.random.default_rng(100 + c)
    n_vox, n_pix, D = 3 * 7 * 11, 5 * 9, 6
    rb, rd, rf, st, ln = random_tables(rng, n_vox, n_pix, n_pix * D, 1500)
    depth = rng.random((1, 1, D, 5, 9), dtype=np.float32)
    feat = rng.standard_normal((1, 1, 5, 9, c), dtype=np.float32)
    shape = (1, 3, 7, 11, c)
    want = OC.bev_pool_v2_fwd(depth, feat, rd, rf, rb, shape, st, ln)
    got = run_fwd(cuda, depth, feat, rd, rf, rb, shape, st, ln)
    assert np.array_equal(got, want)          # same fma chain, same order -> bit-exact


def test_forward_long_intervals_split_path(cuda):
    """Intervals > 512 points take the workgroup-split path: deterministic, ~1e-6 from the oracle."""
    rng = np.random.default_rng(5)
    c, n_vox, n_pix, D = 64, 50, 400, 59
    rb, rd, rf, st, ln = random_tables(rng, n_vox, n_pix, n_pix * D, 20000, long_interval=6000)
    assert ln.max() > 512
    depth = rng.random((1, 1, D, 20, 20), dtype=np.float32)
    feat = rng.standard_normal((1, 1, 20, 20, c), dtype=np.float32)
    shape = (1, 1, 5, 10, c)
    want = OC.bev_pool_v2_fwd(depth, feat, rd, rf, rb, shape, st, ln)
    got = run_fwd(cuda, depth, feat, rd, rf, rb, shape, st, ln)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-4)
    again = run_fwd(cuda, depth, feat, rd, rf, rb, shape, st, ln)
    assert np.array_equal(got, again)          # run-to-run identical (no atomics)


def test_forward_only_named_rows_are_written_and_empty_tables(cuda):
    from projects.mmdet3d_plugin.ops.bev_pool_v2 import bev_pool_v2_ext as ext
    rng = np.random.default_rng(9)
    rb, rd, rf, st, ln = random_tables(rng, 40, 12, 48, 30)
    depth = t(rng.random((1, 1, 4, 3, 4), dtype=np.float32), cuda)
    feat = t(rng.standard_normal((1, 1, 3, 4, 8), dtype=np.float32), cuda)
    out = torch.full((1, 2, 4, 5, 8), 7.0, device=cuda)
    ext.bev_pool_v2_forward(depth, feat, out, t(rd, cuda), t(rf, cuda), t(rb, cuda), t(ln, cuda), t(st, cuda))
    untouched = np.setdiff1d(np.arange(40), rb)
    assert (out.view(40, 8)[torch.from_numpy(untouched).to(cuda)] == 7.0).all()
    e = torch.empty(0, dtype=torch.int32, device=cuda)
    out2 = torch.zeros(1, 2, 4, 5, 8, device=cuda)
    ext.bev_pool_v2_forward(depth, feat, out2, e, e, e, e, e)          # zero intervals: no-op
    assert out2.abs().sum() == 0


@pytest.mark.parametrize("c", [2, 8, 64, 80])
def test_backward_matches_oracle(cuda, c):
    from projects.mmdet3d_plugin.ops.bev_pool_v2 import bev_pool_v2_ext as ext
    rng = np.random.default_rng(200 + c)
    n_vox, n_pix, D = 2 * 6 * 9, 4 * 8, 7
    rb, rd, rf, st, ln = random_tables(rng, n_vox, n_pix, n_pix * D, 180)
    depth = rng.random((1, 1, D, 4, 8), dtype=np.float32)
    feat = rng.standard_normal((1, 1, 4, 8, c), dtype=np.float32)
    og = rng.standard_normal((1, 2, 6, 9, c), dtype=np.float32)
    brb, brd, brf, bst, bln = O.backward_tables(rb, rd, rf)
    want_dg, want_fg = OC.bev_pool_v2_bwd(og, depth, feat, brd, brf, brb, bst, bln)
    dg = torch.zeros(depth.shape, device=cuda)
    fg = torch.zeros(feat.shape, device=cuda)
    ext.bev_pool_v2_backward(t(og, cuda), dg, fg, t(depth, cuda), t(feat, cuda), t(brd, cuda), t(brf, cuda), t(brb, cuda),
                             t(bln, cuda), t(bst, cuda))
    assert np.array_equal(fg.cpu().numpy(), want_fg)            # same fma chain over points
    np.testing.assert_allclose(dg.cpu().numpy(), want_dg, rtol=1e-5, atol=1e-5)   # channel sum is a lane tree


def test_device_backward_tables_match_oracle(cuda):
    from omnihd_amd import ops
    rng = np.random.default_rng(3)
    rb, rd, rf, st, ln = random_tables(rng, 500, 64, 64 * 9, 400)
    got = ops.backward_tables(t(rb, cuda), t(rd, cuda), t(rf, cuda), 64)
    want = O.backward_tables(rb, rd, rf)
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)


def test_csr_dense_forward_and_planned_autograd(cuda, golden):
    """The fused (dense, CSR) forward + cached-plan autograd path equals the reference-API path."""
    import omnihd_amd
    from projects.mmdet3d_plugin.ops.bev_pool_v2.bev_pool import bev_pool_v2
    B, C, Z, Y, X = golden["g4_bev"].shape
    rb, rd, rf = (t(golden[f"g3_{k}"], cuda) for k in ("ranks_bev", "ranks_depth", "ranks_feat"))
    n_pix = golden["g4_feat"].size // C
    for layout in ("bzyx", "byxz"):
        plan = omnihd_amd.plan_from_tables(rb, rd, rf, (B, Z, Y, X), n_pix, layout=layout)
        depth = t(golden["g4_depth"], cuda).requires_grad_()
        feat = t(golden["g4_feat"], cuda).requires_grad_()
        bev = omnihd_amd.plan.planned_pool(depth, feat, plan)
        assert bev.shape == (B, C, Z, Y, X)
        (bev * t(golden["g4_w"], cuda)).sum().backward()
        np.testing.assert_allclose(bev.detach().cpu().numpy(), golden["g4_bev"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(depth.grad.cpu().numpy(), golden["g4_depth_grad"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(feat.grad.cpu().numpy(), golden["g4_feat_grad"], rtol=1e-5, atol=1e-5)
        if layout == "byxz":   # s2c (cat(unbind(2),1)) is a pure reshape of the channels-last buffer
            s2c = torch.cat(bev.unbind(dim=2), 1)
            assert torch.equal(s2c, bev.permute(0, 2, 1, 3, 4).reshape(B, Z * C, Y, X))


@pytest.mark.parametrize("tile_items", [64, 200, 512, 1000])
@pytest.mark.parametrize("c", [64, 8])
def test_tiled_dense_forward_heavy_tail_and_empty_rows(cuda, tile_items, c):
    """Tile tables (csr_tiles, tile_schedule, tile_descriptors: what the direct forward's schedule is made of) on rows far longer
    than a tile and long runs of empty rows, and the any-channel dense kernel on the same rows against the oracle (which only
    writes named rows into a zero buffer); run-to-run identical.  (The C = 64 kernel of the product on such rows:
    test_direct_forward_on_heavy_tailed_rows.)"""
    from omnihd_amd import ops
    rng = np.random.default_rng(tile_items + c)
    n_rows, n_pix, D = 3000, 600, 20
    # heavy tail: a few rows with thousands of points, many singletons, 60 % empty rows
    rows = np.concatenate([np.full(5000, 7), np.full(2100, 8), np.full(3000, 1500), np.full(700, 2999),
                           rng.choice(n_rows, 400, replace=False).repeat(rng.integers(1, 9, 400))])
    rows = np.sort(rows).astype(np.int32)
    npts = rows.size
    rd = rng.integers(0, n_pix * D, npts).astype(np.int32)
    rf = rng.integers(0, n_pix, npts).astype(np.int32)
    st, ln = O.run_length(rows)
    depth = rng.random((1, 1, D, 20, 30), dtype=np.float32)
    feat = rng.standard_normal((1, 1, 20, 30, c), dtype=np.float32)
    want = OC.bev_pool_v2_fwd(depth, feat, rd, rf, rows, (1, 1, 1, n_rows, c), st, ln).reshape(n_rows, c)
    row_ptr = ops.csr_from_sorted_keys(t(rows, cuda), n_rows)
    assert np.array_equal(np.diff(row_ptr.cpu().numpy()), np.bincount(rows, minlength=n_rows))
    tiles = ops.csr_tiles(row_ptr, tile_items, 256)
    tr = tiles.cpu().numpy()
    assert tr[0] == 0 and tr[-1] == n_rows and np.all(np.diff(tr) > 0)
    rp = row_ptr.cpu().numpy()
    lens, tpts = np.diff(rp), rp[tr[1:]] - rp[tr[:-1]]
    assert np.all((tpts <= tile_items + 256) | (np.diff(tr) == 1))      # long rows are tiles of their own
    assert all(np.diff(tr)[np.searchsorted(tr, r, side="right") - 1] == 1 for r in np.nonzero(lens > 256)[0])
    from omnihd_amd.plan import tile_schedule
    order = tile_schedule(row_ptr, tiles, t(rf, cuda), (20, 30))
    o = order.cpu().numpy()
    assert np.array_equal(np.sort(o[o >= 0]), np.arange(tiles.numel() - 1))     # every tile exactly once
    o2 = tile_schedule(row_ptr, tiles, t(rf, cuda), (20, 30), grid=(1, 10, 20, 15), layout="byxz").cpu().numpy()
    assert np.array_equal(np.sort(o2[o2 >= 0]), np.arange(tiles.numel() - 1))
    d_sched = ops.tile_descriptors(row_ptr, tiles, order)
    dn = d_sched.cpu().numpy()
    assert dn[:, 1].sum() == n_rows and dn[:, 3].sum() == npts
    outs = []
    for _ in range(2):
        out = torch.full((n_rows, c), float("nan"), device=cuda)        # every row must be written
        ops.bev_pool_v2_forward_csr(t(depth, cuda), t(feat, cuda), t(rd, cuda), t(rf, cuda), row_ptr, out)
        outs.append(out.cpu().numpy())
    np.testing.assert_allclose(outs[0], want, rtol=1e-5, atol=2e-4)
    assert np.array_equal(outs[0], outs[1])          # run-to-run identical


def test_tiled_dense_forward_all_rows_empty_and_single_row(cuda):
    from omnihd_amd import ops
    depth = torch.rand(1, 1, 2, 2, 2, device=cuda)
    feat = torch.randn(1, 1, 2, 2, 64, device=cuda)
    e = torch.empty(0, dtype=torch.int32, device=cuda)
    row_ptr = ops.csr_from_sorted_keys(e, 1000)
    out = torch.full((1000, 64), 3.0, device=cuda)
    ops.bev_pool_v2_forward_csr(depth, feat, e, e, row_ptr, out)
    assert out.abs().sum() == 0
    rows = torch.zeros(5000, dtype=torch.int32, device=cuda)             # one row holds everything
    g = torch.Generator(device="cpu").manual_seed(11)
    rd = torch.randint(0, 8, (5000,), dtype=torch.int32, generator=g).to(cuda)
    rf = torch.randint(0, 4, (5000,), dtype=torch.int32, generator=g).to(cuda)
    row_ptr = ops.csr_from_sorted_keys(rows, 1)
    out = torch.empty(1, 64, device=cuda)
    ops.bev_pool_v2_forward_csr(depth, feat, rd, rf, row_ptr, out)
    terms = depth.view(-1)[rd.long()][:, None].double() * feat.view(4, 64)[rf.long()].double()
    want = terms.sum(0)
    # 5000 fp32 products summed in the kernel's split order: error bound relative to the sum of magnitudes
    assert float(((out[0].double() - want).abs() / terms.abs().sum(0)).max()) < 2e-6


def test_against_reference_kernels_compiled_by_hipcc(cuda):
    """oracle/_ref = the reference's bev_pool_cuda.cu compiled unmodified for gfx950."""
    ref = RefKernels()
    if not ref.ok:
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    from projects.mmdet3d_plugin.ops.bev_pool_v2 import bev_pool_v2_ext as ext
    rng = np.random.default_rng(77)
    c, n_vox, n_pix, D = 64, 16 * 20 * 30, 6 * 16 * 44, 59
    rb, rd, rf, st, ln = random_tables(rng, n_vox, n_pix, n_pix * D, 120000)
    depth = t(rng.random((1, 6, D, 16, 44), dtype=np.float32), cuda)
    feat = t(rng.standard_normal((1, 6, 16, 44, c), dtype=np.float32), cuda)
    trb, trd, trf, tst, tln = (t(a, cuda) for a in (rb, rd, rf, st, ln))
    ours = torch.zeros(1, 16, 20, 30, c, device=cuda)
    theirs = torch.zeros_like(ours)
    ext.bev_pool_v2_forward(depth, feat, ours, trd, trf, trb, tln, tst)
    ref.v2_fwd(depth, feat, trd, trf, trb, tst, tln, theirs)
    assert torch.equal(ours, theirs)                           # bit-exact vs the reference kernel
    og = t(rng.standard_normal((1, 16, 20, 30, c), dtype=np.float32), cuda)
    brb, brd, brf, bst, bln = (t(a, cuda) for a in O.backward_tables(rb, rd, rf))
    dg, fg = torch.zeros_like(depth), torch.zeros_like(feat)
    dg_r, fg_r = torch.zeros_like(depth), torch.zeros_like(feat)
    ext.bev_pool_v2_backward(og, dg, fg, depth, feat, brd, brf, brb, bln, bst)
    ref.v2_bwd(og, depth, feat, brd, brf, brb, bst, bln, dg_r, fg_r)
    assert torch.equal(fg, fg_r)
    torch.testing.assert_close(dg, dg_r, rtol=1e-5, atol=1e-5)


def test_v1_pool_matches_oracle(cuda):
    from projects.mmdet3d_plugin.ops.bev_pool import bev_pool
    rng = np.random.default_rng(21)
    B, D, H, W, C, n = 2, 3, 5, 7, 16, 900
    coords = np.stack([rng.integers(0, H, n), rng.integers(0, W, n), rng.integers(0, D, n), rng.integers(0, B, n)], 1)
    feats = rng.standard_normal((n, C), dtype=np.float32)
    x = t(feats, cuda).requires_grad_()
    out = bev_pool(x, t(coords, cuda), B, D, H, W)
    assert out.shape == (B, C, D, H, W)
    # oracle on the same (stable) order
    ranks = coords[:, 0] * (W * D * B) + coords[:, 1] * (D * B) + coords[:, 2] * B + coords[:, 3]
    order = np.argsort(ranks, kind="stable")
    st, ln = O.run_length(ranks[order])
    want = OC.bev_pool_v1_fwd(feats[order], coords[order].astype(np.int32), st, ln, B, D, H, W)
    assert np.array_equal(out.detach().cpu().numpy(), want.transpose(0, 4, 1, 2, 3))
    w = rng.standard_normal(out.shape, dtype=np.float32)
    (out * t(w, cuda)).sum().backward()
    xg_sorted = OC.bev_pool_v1_bwd(np.ascontiguousarray(w.transpose(0, 2, 3, 4, 1)), coords[order].astype(np.int32), st, ln, B, D, H, W)
    want_xg = np.empty_like(xg_sorted)
    want_xg[order] = xg_sorted
    assert np.array_equal(x.grad.cpu().numpy(), want_xg)


FULL = {"r1": (64, 176, 2025022), "r2": (136, 240, 4503872)}      # fH, fW, points the reference keeps


@pytest.mark.parametrize("res", ["r1", "r2"])
def test_default_dense_forward_full_size_against_the_oracle(cuda, golden, res):
    """The headline kernel (what ``planned_pool`` launches: k_pool_fwd_direct) at the BASELINE frame size R1 and at the repo's own resolution R2 (544x960) against the CPU
    restatement of the reference kernel on the reference-format tables of the same geometry: every output row, 1e-5 relative
    (summation order of rows cut inside a tile differs; north_star allows 1e-3), empty rows exactly zero, rows with a single
    point bit-exact; the kept-buffer path (rows without points keep their zeros) gives the same bits on its second use."""
    from omnihd_amd import build_plan
    from omnihd_amd.plan import planned_pool
    fH, fW, n_ref = FULL[res]
    geom, dx, bx, nx = full_size_geometry(res)
    plan = build_plan(t(geom, cuda), dx, bx, nx, layout="bzyx")              # the reference's (B,Z,Y,X,C) row order
    rng = np.random.default_rng(5)
    depth = rng.random((1, 6, 59, fH, fW), dtype=np.float32)
    depth /= depth.sum(2, keepdims=True)
    feat = rng.standard_normal((1, 6, fH, fW, 64), dtype=np.float32)
    rb, rd, rf, st, ln = O.voxel_pooling_prepare_v2(geom, dx, bx, nx)
    assert len(rb) == int(golden[f"full_{res}_checksums"][0]) == n_ref     # the point count the reference produced
    want = OC.bev_pool_v2_fwd(depth, feat, rd, rf, rb, (1, 16, 160, 240, 64), st, ln, threads=True)   # (B,Z,Y,X,C)
    scale = float(np.abs(want).max())
    single = np.zeros(16 * 160 * 240, dtype=bool)
    single[rb[st[ln == 1]]] = True
    got = planned_pool(t(depth, cuda), t(feat, cuda), plan)               # logical (B,C,Z,Y,X)
    got = got.permute(0, 2, 3, 4, 1).contiguous().cpu().numpy()
    assert got.shape == want.shape
    assert float(np.abs(got - want).max()) <= 1e-5 * scale
    assert np.array_equal((got == 0).all(-1), (want == 0).all(-1))
    assert np.array_equal(got.reshape(-1, 64)[single], want.reshape(-1, 64)[single])
    for _ in range(3):                                                        # kept buffers: fresh, reused, reused
        again = planned_pool(t(depth, cuda), t(feat, cuda), plan, keep_empty_rows=True)
        again = again.permute(0, 2, 3, 4, 1).contiguous().cpu().numpy()
        assert np.array_equal(again, got)
        del again


@pytest.mark.parametrize("tile_items,long_len", [(64, 64), (200, 256), (768, 512), (3000, 4000)])
@pytest.mark.parametrize("B", [1, 2])
def test_direct_forward_on_heavy_tailed_rows(cuda, tile_items, long_len, B):
    """k_pool_fwd_direct (C = 64) on rows longer than a tile, single-row tiles, empty runs, tiles with fewer than 16 points and
    rows cut at every piece boundary, written into NaN-filled buffers: equals the oracle to 1e-5 (rows of one point bit-exact),
    agrees with the any-channel kernel to the last-bit association difference, is run-to-run identical, does not depend on the
    schedule order, and with ``empty_rows_kept`` leaves exactly the empty rows untouched."""
    from omnihd_amd import ops
    from omnihd_amd.plan import direct_tables_from, tile_schedule
    rng = np.random.default_rng(tile_items + B)
    N, D, fH, fW, c = 2, 7, 5, 12, 64
    fhw, n_rows = fH * fW, 2500
    rows = np.concatenate([np.full(4000, 3), np.full(1300, 4), np.full(2600, 1200), np.full(600, n_rows - 1), np.full(9, 2000),
                           rng.choice(n_rows, 500, replace=False).repeat(rng.integers(1, 12, 500))])
    rows = np.sort(rows).astype(np.int32)
    rd = rng.integers(0, B * N * D * fhw, rows.size).astype(np.int32)
    rf = ((rd // (D * fhw)) * fhw + rd % fhw).astype(np.int32)
    depth = t(rng.random((B, N, D, fH, fW), dtype=np.float32), cuda)
    feat = t(rng.standard_normal((B, N, fH, fW, c), dtype=np.float32), cuda)
    row_ptr = ops.csr_from_sorted_keys(t(rows, cuda), n_rows)
    tiles = ops.csr_tiles(row_ptr, tile_items, long_len)
    st, ln = O.run_length(rows)
    want = OC.bev_pool_v2_fwd(depth.cpu().numpy(), feat.cpu().numpy(), rd, rf, rows, (1, 1, 1, n_rows, c), st, ln).reshape(n_rows, c)
    one = np.bincount(rows, minlength=n_rows) == 1
    empty = np.bincount(rows, minlength=n_rows) == 0
    first = None
    for order in (None, tile_schedule(row_ptr, tiles, t(rf, cuda), (fH, fW))):
        desc = ops.tile_descriptors(row_ptr, tiles, order)
        pt, ivl_rel, desc32 = direct_tables_from(t(rows, cuda), t(rd, cuda), tiles, desc)
        a = torch.full((n_rows, c), float("nan"), device=cuda)
        ops.bev_pool_v2_forward_direct(depth, feat, pt, ivl_rel, desc32, row_ptr, a, D, fhw)
        assert not torch.isnan(a).any()
        np.testing.assert_allclose(a.cpu().numpy(), want, rtol=1e-5, atol=2e-4)
        assert np.array_equal(a.cpu().numpy()[one], want[one])
        assert not a.cpu().numpy()[empty].any()
        a2 = torch.full((n_rows, c), float("nan"), device=cuda)
        ops.bev_pool_v2_forward_direct(depth, feat, pt, ivl_rel, desc32, row_ptr, a2, D, fhw)
        assert torch.equal(a, a2)                                              # run-to-run identical
        first = a if first is None else first
        assert torch.equal(a, first)                                           # the schedule order changes speed only
        k = torch.full((n_rows, c), float("nan"), device=cuda)
        ops.bev_pool_v2_forward_direct(depth, feat, pt, ivl_rel, desc32, row_ptr, k, D, fhw, empty_rows_kept=True)
        kk = k.cpu().numpy()
        assert np.isnan(kk[empty]).all() and np.array_equal(kk[~empty], a.cpu().numpy()[~empty])
        b = torch.full((n_rows, c), float("nan"), device=cuda)                 # the any-channel kernel: row sums in table order
        ops.bev_pool_v2_forward_csr(depth, feat, t(rd, cuda), t(rf, cuda), row_ptr, b)
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


@pytest.mark.parametrize("fH,fW,B", [(5, 12, 1), (8, 12, 2), (4, 44, 1), (3, 7, 2)])
def test_patch_backward_matches_oracle(cuda, fH, fW, B):
    """k_pool_bwd_patch (C = 64): both gradients written densely into NaN-filled buffers, patches that are cut by the end
    of an image (fH*fW not a multiple of 16), pixels without points, against the CPU restatement of the reference kernel:
    feat_grad bit-exact (same fma chain), depth_grad 1e-5 (channel sums in lane order), untouched entries exactly zero."""
    from omnihd_amd import ops
    from omnihd_amd.plan import patch_schedule
    rng = np.random.default_rng(fH * 100 + fW + B)
    N, D, c, n_rows = 2, 7, 64, 3000
    fhw = fH * fW
    n_depth = B * N * D * fhw
    rd = np.sort(rng.permutation(n_depth)[:int(0.6 * n_depth)]).astype(np.int32)        # every frustum point at most once
    rd = rd[(rd // fhw) % D != 3]                                                       # one depth bin never used
    rd = rd[rd % fhw != 5]                                                              # one pixel column never used
    rf = ((rd // (D * fhw)) * fhw + rd % fhw).astype(np.int32)
    rows = rng.integers(0, n_rows, rd.size).astype(np.int32)
    order = np.lexsort((rd, rows))
    brb, brd, brf, bst, bln = O.backward_tables(rows[order], rd[order], rf[order])
    depth = rng.random((B, N, D, fH, fW), dtype=np.float32)
    feat = rng.standard_normal((B, N, fH, fW, c), dtype=np.float32)
    og = rng.standard_normal((1, 1, 1, n_rows, c), dtype=np.float32)
    want_dg, want_fg = OC.bev_pool_v2_bwd(og, depth, feat, brd, brf, brb, bst, bln)
    pix_ptr = ops.csr_from_sorted_keys(t(brf, cuda), B * N * fhw)
    sched = patch_schedule(B * N, (fH, fW))
    assert sorted(sched[sched >= 0].tolist()) == list(range(B * N * ((fhw + 15) // 16)))
    dg = torch.full((B, N, D, fH, fW), float("nan"), device=cuda)
    fg = torch.full((B, N, fH, fW, c), float("nan"), device=cuda)
    ops.bev_pool_v2_backward_patch(t(og.reshape(n_rows, c), cuda), t(depth, cuda), t(feat, cuda), t(brd, cuda), t(brb, cuda),
                                   pix_ptr, sched.to(cuda), dg, fg)
    assert not torch.isnan(dg).any() and not torch.isnan(fg).any()
    assert np.array_equal(fg.cpu().numpy(), want_fg)
    np.testing.assert_allclose(dg.cpu().numpy(), want_dg, rtol=1e-5, atol=1e-5)
    assert np.array_equal(dg.cpu().numpy() == 0, want_dg == 0)
    dg2, fg2 = torch.empty_like(dg), torch.empty_like(fg)
    ops.bev_pool_v2_backward_patch(t(og.reshape(n_rows, c), cuda), t(depth, cuda), t(feat, cuda), t(brd, cuda), t(brb, cuda),
                                   pix_ptr, sched.to(cuda), dg2, fg2)
    assert torch.equal(dg, dg2) and torch.equal(fg, fg2)                                # run-to-run identical
    # the cost-balanced, heaviest-first schedule (what the plan hands to the kernel): same bits
    sched2 = patch_schedule(B * N, (fH, fW), pix_ptr=pix_ptr).to(cuda)
    assert sorted(sched2[sched2 >= 0].tolist()) == sorted(sched[sched >= 0].tolist())
    dg3 = torch.full_like(dg, float("nan")); fg3 = torch.full_like(fg, float("nan"))
    ops.bev_pool_v2_backward_patch(t(og.reshape(n_rows, c), cuda), t(depth, cuda), t(feat, cuda), t(brd, cuda), t(brb, cuda),
                                   pix_ptr, sched2, dg3, fg3)
    assert torch.equal(dg, dg3) and torch.equal(fg, fg3)


@pytest.mark.parametrize("res", ["r1", "r2"])
def test_patch_backward_full_size_against_the_reference_api_kernel(cuda, res):
    """R1 and R2 (the repo's own 544x960) frame geometry through the autograd path of ``planned_pool`` (patch backward) against
    the reference-API backward kernel (itself bit-identical to the reference's own kernel compiled by hipcc) on the plan's
    backward tables."""
    from omnihd_amd import build_plan, ops
    from omnihd_amd.plan import planned_pool
    fH, fW, _ = FULL[res]
    geom, dx, bx, nx = full_size_geometry(res)
    plan = build_plan(t(geom, cuda), dx, bx, nx, layout="byxz")
    assert plan.patch_order is not None and plan.pix_ptr.numel() == 6 * fH * fW + 1
    g = torch.Generator(device="cpu").manual_seed(3)
    depth = torch.rand(1, 6, 59, fH, fW, generator=g).softmax(2).to(cuda).requires_grad_()
    feat = torch.randn(1, 6, fH, fW, 64, generator=g).to(cuda).requires_grad_()
    og = torch.randn(plan.n_rows, 64, generator=g).to(cuda)
    out = planned_pool(depth, feat, plan)                                               # (B,C,Z,Y,X) view of (B,Y,X,Z,C)
    out.backward(og.view(1, 160, 240, 16, 64).permute(0, 4, 3, 1, 2))
    dg, fg = torch.zeros_like(depth), torch.zeros_like(feat)
    ops.bev_pool_v2_backward(og.view(1, 1, 1, plan.n_rows, 64), dg, fg, depth.detach(), feat.detach(), plan.bp_ranks_depth,
                             plan.bp_ranks_feat, plan.bp_ranks_row, plan.bp_lengths, plan.bp_starts)
    assert torch.equal(feat.grad, fg)
    assert float((depth.grad - dg).abs().max()) <= 1e-5 * float(dg.abs().max())
    assert torch.equal(depth.grad == 0, dg == 0)


def test_patch_backward_full_size_r1_directly_against_the_oracle(cuda):
    """the headline backward kernel (k_pool_bwd_patch through ``planned_pool``'s autograd path) at the
    full R1 frame size DIRECTLY against the CPU restatement of the reference kernel (oracle/bev_pool_oracle.c, following
    ops/bev_pool_v2/src/bev_pool_cuda.cu:67-121) on the reference's own backward tables (re-sort by ranks_feat,
    ops/bev_pool_v2/bev_pool.py:47-57) — not through this library's reference-API kernel: feat_grad bit-exact (the same fma
    chain per channel in table order), depth_grad 1e-5 (fixed-order channel sum), the same zero pattern."""
    from omnihd_amd import build_plan
    from omnihd_amd.plan import planned_pool
    fH, fW, n_ref = FULL["r1"]
    geom, dx, bx, nx = full_size_geometry("r1")
    plan = build_plan(t(geom, cuda), dx, bx, nx, layout="bzyx")               # the reference's (B,Z,Y,X,C) row order
    assert plan.patch_order is not None                                       # the patch kernel is what the autograd path launches
    rng = np.random.default_rng(11)
    depth = rng.random((1, 6, 59, fH, fW), dtype=np.float32)
    depth /= depth.sum(2, keepdims=True)
    feat = rng.standard_normal((1, 6, fH, fW, 64), dtype=np.float32)
    og = rng.standard_normal((1, 16, 160, 240, 64), dtype=np.float32)         # (B,Z,Y,X,C)
    rb, rd, rf, st, ln = O.voxel_pooling_prepare_v2(geom, dx, bx, nx)
    assert len(rb) == n_ref
    brb, brd, brf, bst, bln = O.backward_tables(rb, rd, rf)
    want_dg, want_fg = OC.bev_pool_v2_bwd(og, depth, feat, brd, brf, brb, bst, bln, threads=True)
    d = t(depth, cuda).requires_grad_()
    f = t(feat, cuda).requires_grad_()
    out = planned_pool(d, f, plan)                                            # logical (B,C,Z,Y,X)
    out.backward(t(og, cuda).permute(0, 4, 1, 2, 3))
    got_dg, got_fg = d.grad.cpu().numpy(), f.grad.cpu().numpy()
    assert np.array_equal(got_fg, want_fg)
    assert float(np.abs(got_dg - want_dg).max()) <= 1e-5 * float(np.abs(want_dg).max())
    assert np.array_equal(got_dg == 0, want_dg == 0)


def test_kept_output_buffers_survive_consumers_that_write_in_place(cuda, monkeypatch):
    """``planned_pool(keep_empty_rows=True)`` reuses an output buffer whose empty rows are zero already.
    A consumer that writes into the result in place must never make a later forward silently wrong:
      * without autograd the write moves the kept tensor's version counter -> the buffer is zero-filled again before reuse
        (warning, correct result);
      * under autograd torch refuses the write (the result is a view created inside the autograd function);
      * a writer that bypasses torch (an alias with its own version counter stands in for a foreign kernel) is caught by
        OMNIHD_POOL_VERIFY_ZEROS=1;
      * the bytes kept by all plans are bounded (OMNIHD_POOL_KEEP_MAX_MB): beyond the bound the plain path runs."""
    import warnings
    from omnihd_amd import build_plan
    from omnihd_amd import plan as P
    rng = np.random.default_rng(11)
    fr = O.create_frustum((32, 48), 4, [1.0, 9.0, 1.0])
    l2i = O.synthetic_rig(32, 48, 30.0, yaws_deg=(0, 120, -120), radius=0.5, height=0.3)
    inv = [np.linalg.inv(m).astype(np.float32) for m in l2i]
    geom = O.get_geometry(fr, np.stack([m[:3, :3] for m in inv])[None], np.stack([m[:3, 3] for m in inv])[None])
    dx, bx, nx = O.gen_dx_bx([-8.0, 8.0, 1.0], [-6.0, 6.0, 1.0], [-1.0, 1.0, 1.0])
    plan = build_plan(t(geom, cuda), dx, bx, nx, layout="byxz")
    B, N, D, H, W = geom.shape[:5]
    depth = t(rng.random((B, N, D, H, W), dtype=np.float32), cuda)
    feat = t(rng.standard_normal((B, N, H, W, 64), dtype=np.float32), cuda)
    want = P.planned_pool(depth, feat, plan).clone()                       # plain path: every row written
    empty = (plan.row_ptr[1:] == plan.row_ptr[:-1])
    assert int(empty.sum()) > 0
    with torch.no_grad():
        a = P.planned_pool(depth, feat, plan, keep_empty_rows=True)
        assert torch.equal(a, want) and len(plan._kept_outputs) == 1
        a.add_(1.0)                                                        # in-place write into the result (and the kept buffer)
        del a
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            P._WARNED.clear()
            b = P.planned_pool(depth, feat, plan, keep_empty_rows=True)
        assert torch.equal(b, want), "a kept buffer that was written in place must be zero-filled again, not reused as is"
        assert any("written in place" in str(x.message) for x in w)
        del b
        c = P.planned_pool(depth, feat, plan, keep_empty_rows=True)        # clean reuse afterwards, same buffer
        assert torch.equal(c, want) and len(plan._kept_outputs) == 1
        del c
    d = depth.clone().requires_grad_()
    out = P.planned_pool(d, feat, plan, keep_empty_rows=True)
    with pytest.raises(RuntimeError, match="modified inplace|in-place"):
        out.relu_()
    del out, d
    # a writer behind torch's back: same storage, separate version counter
    keeper = plan._kept_outputs[0].tensor
    alias = torch.empty(0, dtype=torch.float32, device=keeper.device).set_(keeper.untyped_storage(), 0, keeper.shape, keeper.stride())
    alias[torch.nonzero(empty).flatten()[:3]] = 7.0
    del alias
    monkeypatch.setenv("OMNIHD_POOL_VERIFY_ZEROS", "1")
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        P._WARNED.clear()
        e = P.planned_pool(depth, feat, plan, keep_empty_rows=True)
    assert torch.equal(e, want) and any("VERIFY_ZEROS" in str(x.message) for x in w)
    del e
    monkeypatch.delenv("OMNIHD_POOL_VERIFY_ZEROS")
    # bound on the kept bytes: a second plan is refused a buffer, its result is still right
    monkeypatch.setenv("OMNIHD_POOL_KEEP_MAX_MB", "0")
    plan2 = build_plan(t(geom, cuda), dx, bx, nx, layout="byxz")
    with torch.no_grad():
        f = P.planned_pool(depth, feat, plan2, keep_empty_rows=True)
    assert torch.equal(f, want) and not getattr(plan2, "_kept_outputs", [])
