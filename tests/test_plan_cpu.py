"""Host-side planning of the pooling kernels (``omnihd_amd/plan.py``: tile and pixel schedules) on CPU tensors: only the
ORDER of work is chosen there, so the properties to hold are "every unit exactly once", consistency with the tables, and
the balance the XCD mapping relies on."""
import numpy as np
import pytest
import torch

from oracle import lss_oracle as O


def _tables(seed=0):
    rng = np.random.default_rng(seed)
    fr = O.create_frustum((32, 48), 4, [1.0, 9.0, 1.0])
    l2i = O.synthetic_rig(32, 48, 30.0, yaws_deg=(0, 90, 180, 270), radius=0.5, height=0.3)
    inv = [np.linalg.inv(m).astype(np.float32) for m in l2i]
    rots = np.stack([m[:3, :3] for m in inv])[None]
    trans = np.stack([m[:3, 3] for m in inv])[None]
    dx, bx, nx = O.gen_dx_bx([-8.0, 8.0, 1.0], [-6.0, 6.0, 1.0], [-1.0, 1.0, 1.0])
    rb, rd, rf, st, ln = O.voxel_pooling_prepare_v2(O.get_geometry(fr, rots, trans), dx, bx, nx)
    return rb, rd, rf, st, ln, tuple(int(v) for v in nx), rng


def _csr(rb, n_rows):
    counts = np.bincount(rb, minlength=n_rows)
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)


@pytest.mark.parametrize("use_grid", [True, False])
@pytest.mark.parametrize("tile_rows", [7, 40])
def test_tile_schedule_is_a_balanced_permutation(use_grid, tile_rows):
    from omnihd_amd.plan import tile_schedule
    rb, rd, rf, st, ln, (X, Y, Z), _ = _tables()
    n_rows = Z * Y * X                                    # B = 1; rows in (z, y, x) order = ranks_bev
    row_ptr = torch.from_numpy(_csr(rb, n_rows))
    tile_row = torch.arange(0, n_rows + tile_rows, tile_rows).clamp(max=n_rows).int()
    tile_row = torch.unique_consecutive(tile_row)
    n_tiles = tile_row.numel() - 1
    kw = dict(grid=(1, Z, Y, X), layout="bzyx") if use_grid else dict(feat_hw=(8, 12))
    flat = tile_schedule(row_ptr, tile_row, torch.from_numpy(rf), n_xcd=8, **kw)
    per = (n_tiles + 7) // 8
    assert flat.dtype == torch.int32 and flat.numel() == 8 * per
    used = flat[flat >= 0].numpy()
    assert sorted(used.tolist()) == list(range(n_tiles))                     # every tile exactly once
    # work (points + rows) per XCD: balanced up to the largest tile
    lo, hi = row_ptr[tile_row[:-1].long()].numpy(), row_ptr[tile_row[1:].long()].numpy()
    work = (hi - lo) + (tile_row[1:] - tile_row[:-1]).numpy()
    per_xcd = [int(work[flat[k * per:(k + 1) * per][flat[k * per:(k + 1) * per] >= 0].numpy()].sum()) for k in range(8)]
    assert max(per_xcd) - min(per_xcd) <= 2 * int(work.max()) + int(work.sum()) // 8 // 4, per_xcd
    if use_grid:                                                              # azimuth order: a run is an angular wedge
        mid = ((tile_row[:-1] + tile_row[1:]) // 2).long().clamp(max=n_rows - 1).numpy()
        yy, xx = (mid // X) % Y, mid % X
        ang = np.arctan2(yy - (Y - 1) / 2.0, xx - (X - 1) / 2.0)
        first = flat[:per][flat[:per] >= 0].numpy()
        assert np.all(np.diff(ang[first]) >= -1e-12)


def test_tile_schedule_spills_when_one_run_holds_too_many_tiles():
    """Many empty tiles sort together: a run longer than its slot count spills into free slots, nothing is lost."""
    from omnihd_amd.plan import tile_schedule
    n_rows, tile_rows = 640, 4
    counts = np.zeros(n_rows, dtype=np.int64)
    counts[:16] = 500                                       # all the points in the first four tiles
    row_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32))
    tile_row = torch.arange(0, n_rows + 1, tile_rows).int()
    rf = torch.zeros(int(counts.sum()), dtype=torch.int32)
    flat = tile_schedule(row_ptr, tile_row, rf, feat_hw=None, n_xcd=8)
    n_tiles = tile_row.numel() - 1
    assert sorted(flat[flat >= 0].tolist()) == list(range(n_tiles)) and int((flat < 0).sum()) == flat.numel() - n_tiles


def test_patch_schedule_lists_every_patch_once():
    from omnihd_amd.plan import patch_schedule
    for n_img, hw in ((6, (64, 176)), (3, (8, 12)), (2, (5, 12)), (1, (3, 7))):
        sched = patch_schedule(n_img, hw)
        ppi = (hw[0] * hw[1] + 15) // 16
        assert sched.dtype == torch.int32 and sched.numel() % 8 == 0
        assert sorted(sched[sched >= 0].tolist()) == list(range(n_img * ppi))
        per = sched.numel() // 8
        counts = [(sched[k * per:(k + 1) * per] >= 0).sum().item() for k in range(8)]
        assert max(counts) - min(counts) <= 1                                           # equal work per XCD


def test_cost_balanced_patch_schedule_is_a_permutation_with_equal_cost_runs():
    """With the points per pixel the runs of the patch schedule carry (nearly) equal cost and start with their heaviest
    4-row blocks; still every patch exactly once."""
    from omnihd_amd.plan import PATCH, PATCH_FIXED_COST, patch_schedule
    g = torch.Generator().manual_seed(3)
    n_img, (fH, fW) = 6, (64, 176)
    fhw = fH * fW
    # points per pixel: heavy band in the middle rows of every image, nothing at the top
    rows = torch.arange(fH).view(1, fH, 1).expand(n_img, fH, fW)
    lens = (59.0 * torch.exp(-((rows - 40.0) / 12.0) ** 2)).long() + torch.randint(0, 3, (n_img, fH, fW), generator=g)
    lens[:, :8] = 0
    pix_ptr = torch.cat([torch.zeros(1, dtype=torch.long), lens.view(-1).cumsum(0)]).int()
    sched = patch_schedule(n_img, (fH, fW), pix_ptr=pix_ptr)
    ppi = fhw // PATCH
    live = sched[sched >= 0]
    assert sched.numel() % 8 == 0 and sorted(live.tolist()) == list(range(n_img * ppi))
    cost = lens.view(n_img, ppi, PATCH).sum(-1).view(-1) + PATCH_FIXED_COST
    per = sched.numel() // 8
    run_cost = []
    for k in range(8):
        run = sched[k * per:(k + 1) * per]
        run = run[run >= 0].long()
        assert run.numel() > 0
        run_cost.append(float(cost[run].sum()))
        # heaviest first: the first quarter of a run costs more per patch than its last quarter
        q = max(1, run.numel() // 4)
        assert float(cost[run[:q]].float().mean()) >= float(cost[run[-q:]].float().mean())
    assert max(run_cost) <= 1.03 * min(run_cost)
    # the count-balanced schedule of the same frame is worse
    flat = patch_schedule(n_img, (fH, fW))
    per0 = flat.numel() // 8
    c0 = [float(cost[flat[k * per0:(k + 1) * per0][flat[k * per0:(k + 1) * per0] >= 0].long()].sum()) for k in range(8)]
    assert max(c0) / min(c0) > max(run_cost) / min(run_cost)


def test_packed_row_and_depth_bin_table_of_the_patch_backward(monkeypatch):
    """plan._row_bin: one int32 per frustum point = output row | depth bin << 24 (what omnihd_bev_pool_v2_bwd_patch reads when
    ranks_depth is NULL); the bin is recovered from the depth rank as (rank // (fH*fW)) % D; limits fall back to two tables."""
    from types import SimpleNamespace
    from omnihd_amd.plan import _row_bin
    rng = np.random.default_rng(3)
    n_img, D, fhw, n_rows = 5, 59, 7 * 11, 4000
    img, d, hw = rng.integers(0, n_img, 900), rng.integers(0, D, 900), rng.integers(0, fhw, 900)
    rd = torch.from_numpy(((img * D + d) * fhw + hw).astype(np.int32))
    rr = torch.from_numpy(rng.integers(0, n_rows, 900).astype(np.int32))
    plan = SimpleNamespace(bp_row_bin=None, depth_bins=D, n_rows=n_rows, feat_hw=fhw, bp_ranks_depth=rd, bp_ranks_row=rr)
    packed = _row_bin(plan)
    assert packed.dtype == torch.int32 and packed is plan.bp_row_bin
    assert torch.equal(packed & 0x00ffffff, rr) and torch.equal((packed >> 24) & 0xff, torch.from_numpy(d.astype(np.int32)))
    assert _row_bin(plan) is packed                                     # built once per plan
    for bad in (dict(depth_bins=128), dict(n_rows=0x00ffffff), dict(depth_bins=0)):
        p2 = SimpleNamespace(**{**vars(plan), "bp_row_bin": None, **bad})
        assert _row_bin(p2) is None
    monkeypatch.setenv("OMNIHD_POOL_BWD_PACKED", "0")
    assert _row_bin(plan) is None


def _walk_direct_tables(depth, feat, pt, ivl_rel, desc32, n_rows, d_bins, fhw):
    """Host emulation of k_pool_fwd_direct's walk (csrc/bev_pool_v2.hip): 16 groups per tile, pieces of ceil(n/16) points,
    closing flags in bit 31 of the point word, output rows from ivl_rel, partials of cut rows combined in piece order."""
    C = feat.shape[-1]
    out = np.zeros((n_rows, C), dtype=np.float64)
    written = np.zeros(n_rows, dtype=np.int64)
    dflat, frows = depth.reshape(-1).astype(np.float64), feat.reshape(-1, C).astype(np.float64)
    for s in range(desc32.shape[0]):
        Ra, nrows, Pa, npts = (int(v) for v in desc32[s, :4])
        if nrows <= 0 or npts == 0:
            continue
        w = (npts + 15) // 16
        tails, flags, heads, closed_own = [], [], {}, []
        for g in range(16):
            q0, q1 = Pa + min(g * w, npts), Pa + min(g * w + w, npts)
            gi = int(desc32[s, 8 + g])
            ivl, pend = gi & 0x7fffffff, gi < 0 and q0 < q1
            was = pend
            acc = np.zeros(C)
            for q in range(q0, q1):
                p = int(pt[q])
                rd = p & 0x7fffffff
                pix = (rd // (d_bins * fhw)) * fhw + rd % fhw
                acc = acc + dflat[rd] * frows[pix]
                if p < 0:
                    row = Ra + int(ivl_rel[ivl]); ivl += 1
                    assert Ra <= row < Ra + nrows
                    if pend:
                        heads[g] = (acc, row); pend = False
                    else:
                        out[row] = acc; written[row] += 1
                    acc = np.zeros(C)
            open_end = q1 <= q0 or int(pt[q1 - 1]) >= 0
            tails.append(acc); flags.append((1 if open_end else 0) | (2 if (pend or q1 <= q0) else 0)); closed_own.append(was and not pend)
        for g in range(16):
            if closed_own[g]:
                g0 = g
                while g0 > 0:
                    f = flags[g0 - 1]
                    if not f & 1:
                        break
                    g0 -= 1
                    if not f & 2:
                        break
                acc, row = heads[g]
                out[row] = sum(tails[g0:g], np.zeros(C)) + acc; written[row] += 1
    return out, written


@pytest.mark.parametrize("tile_rows", [3, 17, 64])
def test_direct_forward_tables_walk_every_point_once_and_close_every_row(tile_rows):
    """plan.direct_tables_from: the per-point word, the row table and the 32-int tile descriptors drive a walk that equals the
    pooling oracle on every row (tiny rig; tiles so small that most rows are cut by a piece boundary, and single-row tiles)."""
    from omnihd_amd.plan import direct_tables_from
    from oracle import cpu as OC
    rb, rd, rf, st, ln, (X, Y, Z), rng = _tables()
    n_rows = Z * Y * X
    row_ptr = _csr(rb, n_rows)
    tile_row = np.unique(np.concatenate([np.arange(0, n_rows, tile_rows), [n_rows]])).astype(np.int32)
    n_tiles = len(tile_row) - 1
    S = 8 * ((n_tiles + 7) // 8)
    order = rng.permutation(S)                                   # any schedule order: results must not depend on it
    desc = np.zeros((S, 4), dtype=np.int32)
    for t in range(n_tiles):
        ra, rb_ = tile_row[t], tile_row[t + 1]
        desc[order[t]] = (ra, rb_ - ra, row_ptr[ra], row_ptr[rb_] - row_ptr[ra])
    pt, ivl_rel, desc32 = direct_tables_from(torch.from_numpy(rb.astype(np.int32)), torch.from_numpy(rd), torch.from_numpy(tile_row),
                                             torch.from_numpy(desc))
    assert pt.dtype == torch.int32 and desc32.shape == (S, 32) and ivl_rel.numel() == len(st)
    N, D, H, W, C = 4, 8, 8, 12, 8
    depth = rng.random((1, N, D, H, W), dtype=np.float32)
    feat = rng.standard_normal((1, N, H, W, C), dtype=np.float32)
    got, written = _walk_direct_tables(depth, feat, pt.numpy(), ivl_rel.numpy(), desc32.numpy(), n_rows, D, H * W)
    want = OC.bev_pool_v2_fwd(depth, feat, rd, rf, rb, (1, Z, Y, X, C), st, ln).reshape(n_rows, C)
    nonempty = np.diff(row_ptr) > 0
    assert np.array_equal(written > 0, nonempty) and written.max() == 1      # every non-empty row written exactly once
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n_img,fH,fW", [(6, 64, 176), (12, 136, 240), (2, 5, 24), (3, 7, 16)])
def test_static_patch_walk_of_the_device_plan_is_the_walk_of_the_host_schedule(n_img, fH, fW):
    """omnihd_amd.pool_plan.patch_walk (what csrc/pool_plan.hip cuts into XCD runs on the device) lists every patch once, band by
    band, and without per-pixel costs the host's patch_schedule is exactly this walk cut into 8 equal runs."""
    from omnihd_amd import plan as P, pool_plan
    walk, band = pool_plan.patch_walk(n_img, fH, fW)
    ppi = (fH * fW + 15) // 16
    assert sorted(walk.tolist()) == list(range(n_img * ppi))
    assert bool((band[1:] >= band[:-1]).all()), "bands in ascending order"
    flat = P.patch_schedule(n_img, (fH, fW)).view(8, -1)
    assert [p for run in flat.tolist() for p in run if p >= 0] == walk.tolist()
