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


def test_pixel_schedule_lists_every_pixel_once_with_its_points():
    from omnihd_amd.plan import pixel_schedule
    rb, rd, rf, st, ln, _, _ = _tables()
    bp = O.backward_tables(rb, rd, rf)                     # (ranks_bev, ranks_depth, ranks_feat, starts, lengths) sorted by pixel
    n_pix = 4 * 8 * 12
    desc = pixel_schedule(torch.from_numpy(bp[2]), torch.from_numpy(bp[3]), torch.from_numpy(bp[4]), n_pix, feat_hw=(8, 12))
    per = (n_pix + 7) // 8
    assert desc.shape == (8 * per, 4) and desc.dtype == torch.int32
    rows = desc[desc[:, 0] >= 0]
    assert sorted(rows[:, 0].tolist()) == list(range(n_pix))                 # pixels without points are listed too
    starts, lengths = bp[3], bp[4]
    for pix, s, n, z in rows.tolist():
        assert z == 0
        if n:
            assert np.all(bp[2][s:s + n] == pix)                              # its run of the pixel-sorted tables
        else:
            assert s == 0
    assert int(rows[:, 2].sum()) == len(rb)
    # 4x4 patches: the first 16 descriptors of a run cover one 4x4 block of one image
    first = rows[:16, 0].numpy()
    h, w = (first // 12) % 8, first % 12
    assert len({int(v) for v in first // 96}) == 1 and h.max() - h.min() <= 3 and w.max() - w.min() <= 3
    empty = pixel_schedule(torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int32),
                           10, feat_hw=None)
    assert sorted(empty[empty[:, 0] >= 0][:, 0].tolist()) == list(range(10)) and int(empty[:, 2].sum()) == 0


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


def _walk_stream_tables(out_grad, depth, feat, tb, d_bins, fh, fw):
    """Host emulation of k_pool_bwd_stream's walk (csrc/bev_pool_v2.hip), iteration by iteration as the kernel does it: entry t
    names the stage consumed now (t-2), the stage whose rows are fetched now was named by entry t-1, the stage whose ids / offsets
    are read now is entry t's own; the pixel data of a patch are fetched one iteration before its first stage."""
    C = feat.shape[-1]
    og = out_grad.reshape(-1, C).astype(np.float64)
    dflat = depth.reshape(-1).astype(np.float64)
    frows = feat.reshape(-1, C).astype(np.float64)
    dg = np.full(dflat.shape, np.nan)
    fg = np.full(frows.shape, np.nan)
    pw, R = tb.patch_w, tb.rows_per_stage
    ph = 16 // pw
    pcols, prows = -(-fw // pw), -(-fh // ph)
    fhw = fh * fw
    stream, sptr = tb.stream.numpy().astype(np.int64), tb.stream_ptr.numpy()
    uniq, off, word = tb.uniq_rows.numpy(), tb.px_off.numpy().reshape(-1, 16), tb.pt_word.numpy()
    consumed = []

    def geometry(patch):
        img, pr, pc = patch // (pcols * prows), (patch // pcols) % prows, patch % pcols
        return [(img, (pr * ph + g // pw) * fw + pc * pw + g % pw) if (pr * ph + g // pw < fh and pc * pw + g % pw < fw) else None
                for g in range(16)]

    for w in range(tb.n_streams):
        ids_prev, nrows_prev, o1, o2, rows_lds, rows_regs, geo_next, geo, acc = None, 0, None, None, None, None, None, None, None
        for e in range(sptr[w], sptr[w + 1]):
            ex, ey, ez, ew = (int(v) for v in stream[e])
            rows_lds = rows_regs                                            # (1) the rows requested last iteration -> LDS
            if ex & (1 << 30):
                geo = geo_next
                acc = {g: np.zeros(C) for g in range(16) if geo[g] is not None}
                for g, px in enumerate(geo):
                    if px is not None:
                        dg[(px[0] * d_bins + np.arange(d_bins)) * fhw + px[1]] = 0.0
            ids_n = uniq[ey:ey + 64]                                        # (2) requests
            so, nrows = ez & 0xffffff, (ez >> 24) & 0xff
            o_n = (off[so].copy(), off[so + 1].copy())
            if ew >= 0:
                geo_next = geometry(ew)
            rows_regs = None if ids_prev is None else [ids_prev[r] if r < nrows_prev else None for r in range(R)]
            if ex & (1 << 28):                                              # (3) the points of stage t-2
                assert (ex & 0x0fffffff) >= 0 and geo is not None
                consumed.append((ex & 0x0fffffff, bool(ex & (1 << 30)), bool(ex & (1 << 29))))
                a, b = o2
                for g, px in enumerate(geo):
                    if px is None:
                        assert a[g] == b[g]
                        continue
                    assert 0 <= b[g] - a[g] <= 64
                    f = px[0] * fhw + px[1]
                    for q in range(a[g], b[g]):
                        wd = int(word[q])
                        lid, dk = (wd & 0xffffff) >> 8, (wd >> 24) & 0xff
                        row = og[rows_lds[lid]]
                        rd = (px[0] * d_bins + dk) * fhw + px[1]
                        acc[g] = acc[g] + dflat[rd] * row
                        dg[rd] = float(row @ frows[f])
                if ex & (1 << 29):
                    for g, px in enumerate(geo):
                        if px is not None:
                            fg[px[0] * fhw + px[1]] = acc[g]
            o2, o1 = o1, o_n                                                # (4) rotate
            ids_prev, nrows_prev = ids_n, nrows
    return dg, fg, consumed


@pytest.mark.parametrize("patch_w,rows_per_stage,streams_per_xcd", [(16, 32, 1), (8, 32, 2), (4, 32, 1), (8, 64, 3)])
def test_stream_backward_tables_walk_every_stage_once(patch_w, rows_per_stage, streams_per_xcd):
    """plan.stream_tables_from: the per-wave streams (two entries more than stages, every entry naming the stage consumed, the
    stage whose rows are on their way and the stage whose ids are read) drive a walk that equals the pooling oracle's backward;
    every patch's stages appear once, in order, first/last flagged; patches without points still write their zeros."""
    from omnihd_amd.plan import stream_tables_from
    from oracle import cpu as OC
    rb, rd, rf, st, ln, (X, Y, Z), rng = _tables()
    N, D, H, W, C = 4, 8, 8, 12, 8
    n_rows = Z * Y * X
    keep = (rf // (H * W) != 2) | ((rf % (H * W)) // W >= 4)       # no points in the upper half of image 2: patches without a stage
    rb, rd, rf = rb[keep], rd[keep], rf[keep]
    order = np.lexsort((rb, rf))
    brb, brd, brf = rb[order].astype(np.int32), rd[order].astype(np.int32), rf[order].astype(np.int32)
    pix_ptr = np.concatenate([[0], np.cumsum(np.bincount(brf, minlength=N * H * W))]).astype(np.int32)
    tb = stream_tables_from(torch.from_numpy(brb), torch.from_numpy(brd), torch.from_numpy(pix_ptr), N, D, (H, W), patch_w, rows_per_stage,
                            streams_per_xcd)
    assert tb is not None and tb.n_streams == 8 * streams_per_xcd and tb.stream.shape[1] == 4 and tb.balance >= 1.0
    depth = rng.random((1, N, D, H, W), dtype=np.float32)
    feat = rng.standard_normal((1, N, H, W, C), dtype=np.float32)
    out_grad = rng.standard_normal((n_rows, C), dtype=np.float32)
    dg, fg, consumed = _walk_stream_tables(out_grad, depth, feat, tb, D, H, W)
    n_patch = N * -(-W // patch_w) * -(-H // (16 // patch_w))
    firsts = [p for p, first, last in consumed if first]
    lasts = [p for p, first, last in consumed if last]
    assert sorted(firsts) == list(range(n_patch)) and sorted(lasts) == list(range(n_patch))      # every patch opened and closed once
    assert not np.isnan(dg).any() and not np.isnan(fg).any()
    bst = np.flatnonzero(np.r_[True, brf[1:] != brf[:-1]]).astype(np.int32)
    bln = np.diff(np.r_[bst, len(brf)]).astype(np.int32)
    want_dg, want_fg = OC.bev_pool_v2_bwd(out_grad.reshape(1, Z, Y, X, C), depth, feat, brd, brf, brb, bst, bln)
    np.testing.assert_allclose(dg.reshape(want_dg.shape), want_dg, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(fg.reshape(want_fg.shape), want_fg, rtol=1e-5, atol=1e-6)


def test_stream_backward_tables_refuse_what_the_kernel_cannot_walk():
    from omnihd_amd.plan import stream_tables_from
    rows = torch.tensor([5, 3], dtype=torch.int32)                # one pixel whose two points are NOT sorted by row
    rd = torch.tensor([0, 16], dtype=torch.int32)
    pix_ptr = torch.zeros(17, dtype=torch.int32); pix_ptr[1:] = 2
    assert stream_tables_from(rows, rd, pix_ptr, 1, 2, (4, 4), 4, 32, 1) is None
    ok = stream_tables_from(torch.tensor([3, 5], dtype=torch.int32), rd, pix_ptr, 1, 2, (4, 4), 4, 32, 1)
    assert ok is not None and ok.uniq_rows.tolist() == [3, 5] and ok.stream.shape == (2 * 8 + 1, 4)        # one stage + two entries per wave
    assert stream_tables_from(torch.tensor([3, 5], dtype=torch.int32), rd, pix_ptr, 1, 65, (4, 4), 4, 32, 1) is None     # > 64 depth bins
    assert stream_tables_from(torch.tensor([3, 5], dtype=torch.int32), rd, pix_ptr, 1, 2, (4, 4), 5, 32, 1) is None      # patch width
    assert stream_tables_from(torch.tensor([3, 5], dtype=torch.int32), rd, pix_ptr, 1, 2, (4, 4), 4, 40, 1) is None      # rows per stage



def test_stream_dealing_keeps_the_walk_order_and_balances_the_waves():
    """plan.shared_schedule + plan.stream_deal: every patch of every patch shape lands in exactly one stream, a stream's patches
    keep the walk order of their XCD run (neighbouring patches close in time), and the dealt cost per wave is even (each next
    patch goes to the least loaded wave) although patch costs vary by 10x."""
    from omnihd_amd.plan import shared_schedule, stream_deal
    rng = np.random.default_rng(5)
    n_img, feat_hw = 6, (64, 176)
    for patch_w in (16, 8, 4):
        n_patch = n_img * -(-feat_hw[1] // patch_w) * -(-feat_hw[0] // (16 // patch_w))
        cost = torch.from_numpy(rng.integers(100, 1000, n_patch)).double()
        runs = shared_schedule(n_img, feat_hw, patch_w, cost)
        assert len(runs) == 8 and sorted(torch.cat(runs).tolist()) == list(range(n_patch))
        run_cost = [float(cost[r].sum()) for r in runs]
        assert max(run_cost) / min(run_cost) < 1.02                                      # XCD runs cut by cost
        lists = stream_deal(runs, cost, 24)
        assert len(lists) == 8 * 24 and sorted(p for l in lists for p in l) == list(range(n_patch))
        for x, run in enumerate(runs):
            pos = {int(p): i for i, p in enumerate(run.tolist())}
            for l in lists[x * 24:(x + 1) * 24]:
                assert all(p in pos for p in l) and [pos[p] for p in l] == sorted(pos[p] for p in l)   # same XCD, walk order kept
        load = np.array([float(cost[l].sum()) for l in lists])
        assert load.max() / load.mean() < 1.15


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
