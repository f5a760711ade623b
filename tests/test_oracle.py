"""CPU tests: pin the oracle against the reference's known-answer test and the golden vectors
captured from the reference Python (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import cpu as OC
from oracle import lss_oracle as O


def kat_tables():
    # values of the reference's test_bev_pool_v2 (ops/bev_pool_v2/bev_pool.py:145-176)
    depth = np.array([0.3, 0.4, 0.2, 0.1, 0.7, 0.6, 0.8, 0.9], dtype=np.float32).reshape(1, 1, 2, 2, 2)
    feat = np.ones((1, 1, 2, 2, 2), dtype=np.float32)
    rd = np.array([0, 4, 1, 6], dtype=np.int32)
    rf = np.array([0, 0, 1, 2], dtype=np.int32)
    rb = np.array([0, 0, 1, 1], dtype=np.int32)
    st, ln = O.run_length(rb)
    return depth, feat, rd, rf, rb, st, ln


def test_reference_known_answer_forward_backward():
    depth, feat, rd, rf, rb, st, ln = kat_tables()
    out = OC.bev_pool_v2_fwd(depth, feat, rd, rf, rb, (1, 1, 2, 2, 2), st, ln)
    bev = out.transpose(0, 4, 1, 2, 3)                          # bev_pool.py:91
    assert np.float32(bev.sum()) == np.float32(4.4)             # reference :169
    # loss = sum(bev) -> out_grad = ones; backward tables as bev_pool.py:47-57
    brb, brd, brf, bst, bln = O.backward_tables(rb, rd, rf)
    dg, fg = OC.bev_pool_v2_bwd(np.ones_like(out), depth, feat, brd, brf, brb, bst, bln)
    np.testing.assert_allclose(dg.reshape(-1), [2., 2., 0., 0., 2., 0., 2., 0.])            # :170-173
    np.testing.assert_allclose(fg.reshape(-1), [1.0, 1.0, 0.4, 0.4, 0.8, 0.8, 0., 0.], rtol=1e-6)  # :174-176


@pytest.mark.parametrize("tag,final_dim", [("r1", (256, 704)), ("r2", (544, 960)), ("tiny", (32, 48))])
def test_grid_constants_and_frustum(golden, tag, final_dim):
    pc = [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]
    dx, bx, nx = O.gen_dx_bx([pc[0], pc[3], 0.5], [pc[1], pc[4], 0.5], [pc[2], pc[5], 0.5])
    assert np.array_equal(dx, golden[f"g1_{tag}_dx"])
    assert np.array_equal(bx, golden[f"g1_{tag}_bx"])
    assert np.array_equal(nx, golden[f"g1_{tag}_nx"])
    assert nx.tolist() == [240, 160, 16] and bx.tolist() == [-59.75, -39.75, -2.75]
    xs, ys, ds = O.frustum_axes(final_dim, 4, [1, 60, 1])
    assert np.array_equal(xs, golden[f"g1_{tag}_xs"])
    assert np.array_equal(ys, golden[f"g1_{tag}_ys"])
    assert np.array_equal(ds, golden[f"g1_{tag}_ds"])
    assert len(ds) == 59


def _tiny_setup(golden):
    pc = golden["g2_pc_range"].tolist()
    g = float(golden["g2_grid"])
    dx, bx, nx = O.gen_dx_bx([pc[0], pc[3], g], [pc[1], pc[4], g], [pc[2], pc[5], g])
    fr = O.create_frustum(tuple(golden["g2_final_dim"].tolist()), 4, golden["g2_dbound"].tolist())
    return dx, bx, nx, fr


def test_geometry_matches_reference(golden):
    dx, bx, nx, fr = _tiny_setup(golden)
    geom = O.get_geometry(fr, golden["g2_rots"], golden["g2_trans"])
    np.testing.assert_allclose(geom, golden["g2_geom"], rtol=0, atol=1e-5)


def test_prepare_tables_match_reference(golden):
    dx, bx, nx, _ = _tiny_setup(golden)
    tabs = O.voxel_pooling_prepare_v2(golden["g2_geom"], dx, bx, nx)
    for k, t in zip(["ranks_bev", "ranks_depth", "ranks_feat", "starts", "lengths"], tabs):
        assert np.array_equal(t, golden[f"g3_{k}"]), k


def test_prepare_adversarial_matches_reference(golden):
    dx, bx, nx = O.gen_dx_bx([-2.0, 2.0, 1.0], [-2.0, 2.0, 1.0], [-1.0, 1.0, 1.0])
    tabs = O.voxel_pooling_prepare_v2(golden["g3adv_coor"], dx, bx, nx)
    for k, t in zip(["ranks_bev", "ranks_depth", "ranks_feat", "starts", "lengths"], tabs):
        assert np.array_equal(t, golden[f"g3adv_{k}"]), k
    # defect D3: (-2.5,-1.5,-0.5) -> x voxel -0.5 truncates to 0 and is KEPT; NaN / +-1e30 dropped
    assert 0 in tabs[1] and 9 not in tabs[1] and 10 not in tabs[1] and 11 not in tabs[1]


def test_prepare_empty_returns_none():
    dx, bx, nx = O.gen_dx_bx([-2.0, 2.0, 1.0], [-2.0, 2.0, 1.0], [-1.0, 1.0, 1.0])
    coor = np.full((1, 1, 2, 2, 2, 3), 100.0, dtype=np.float32)
    assert O.voxel_pooling_prepare_v2(coor, dx, bx, nx) == (None,) * 5


def test_pool_through_reference_autograd_function(golden):
    """g4_* were produced by the reference's QuickCumsumCuda (python logic) on the tiny rig."""
    st, ln = golden["g3_starts"], golden["g3_lengths"]
    depth, feat = golden["g4_depth"], golden["g4_feat"]
    B, C, Z, Y, X = golden["g4_bev"].shape
    # (1) with the tables in the order the reference itself produced: bit-identical results
    rb, rd, rf = golden["g3raw_ranks_bev"], golden["g3raw_ranks_depth"], golden["g3raw_ranks_feat"]
    out = OC.bev_pool_v2_fwd(depth, feat, rd, rf, rb, (B, Z, Y, X, C), st, ln)
    assert np.array_equal(out.transpose(0, 4, 1, 2, 3), golden["g4_bev"])
    og = np.ascontiguousarray(golden["g4_w"].transpose(0, 2, 3, 4, 1))
    dg, fg = OC.bev_pool_v2_bwd(og, depth, feat, golden["g4raw_bp_ranks_depth"], golden["g4raw_bp_ranks_feat"],
                                golden["g4raw_bp_ranks_bev"], golden["g4_bp_starts"], golden["g4_bp_lengths"])
    assert np.array_equal(dg, golden["g4_depth_grad"])
    assert np.array_equal(fg, golden["g4_feat_grad"])
    # (2) canonical (stable) tables: same intervals, summation order inside an interval differs
    rb, rd, rf = golden["g3_ranks_bev"], golden["g3_ranks_depth"], golden["g3_ranks_feat"]
    out = OC.bev_pool_v2_fwd(depth, feat, rd, rf, rb, (B, Z, Y, X, C), st, ln)
    np.testing.assert_allclose(out.transpose(0, 4, 1, 2, 3), golden["g4_bev"], rtol=1e-5, atol=1e-6)
    bp = O.backward_tables(rb, rd, rf)
    for k, t in zip(["ranks_bev", "ranks_depth", "ranks_feat", "starts", "lengths"], bp):
        assert np.array_equal(t, golden[f"g4_bp_{k}"]), k
    dg, fg = OC.bev_pool_v2_bwd(og, depth, feat, bp[1], bp[2], bp[0], bp[3], bp[4])
    np.testing.assert_allclose(dg, golden["g4_depth_grad"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(fg, golden["g4_feat_grad"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag,H,W,fx", [("r1", 256, 704, 410.0)])
def test_full_size_checksums(golden, tag, H, W, fx):
    """Full-size (R1) tables from the oracle equal the reference's, via checksums."""
    pc = [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]
    dx, bx, nx = O.gen_dx_bx([pc[0], pc[3], 0.5], [pc[1], pc[4], 0.5], [pc[2], pc[5], 0.5])
    fr = O.create_frustum((H, W), 4, [1, 60, 1])
    assert np.array_equal(O.synthetic_rig(H, W, fx), golden[f"full_{tag}_lidar2img"])
    geom = O.get_geometry(fr, golden[f"full_{tag}_rots"], golden[f"full_{tag}_trans"])
    np.testing.assert_allclose(geom.astype(np.float64).sum(axis=(0, 1, 2, 3, 4)), golden[f"full_{tag}_geom_sum"], rtol=1e-9)
    tabs = O.voxel_pooling_prepare_v2(geom, dx, bx, nx)
    cs = [tabs[0].size, tabs[3].size] + [int(t.astype(np.int64).sum()) for t in tabs] + [int(tabs[4].max())]
    assert cs == golden[f"full_{tag}_checksums"].tolist()


# ---- radar side: hand-computed cases (upstream semantics; parity unpinned by the reference) ----
def test_hard_voxelize_first_occurrence_order_and_caps():
    vs, rng = [1.0, 1.0, 4.0], [0.0, 0.0, -2.0, 4.0, 4.0, 2.0]
    pts = np.array([[2.5, 1.5, 0.0, 10],    # voxel (x2,y1) -> id 0
                    [0.5, 0.5, 0.0, 11],    # voxel (0,0)   -> id 1
                    [2.6, 1.4, 1.0, 12],    # id 0, slot 1
                    [9.0, 0.0, 0.0, 13],    # outside
                    [2.7, 1.3, 0.0, 14],    # id 0, slot 2 -> dropped by max_points=2
                    [3.5, 3.5, 0.0, 15],    # voxel (3,3)   -> id 2
                    [0.1, 0.9, -2.0, 16],   # id 1 slot 1 (z = range min is inside)
                    [1.5, 1.5, 2.0, 17],    # z = range max -> outside
                    [1.5, 1.5, 0.0, 18]],   # would be id 3 -> refused by max_voxels=3
                   dtype=np.float32)
    v, c, n = OC.hard_voxelize(pts, vs, rng, max_points=2, max_voxels=3)
    assert c.tolist() == [[0, 1, 2], [0, 0, 0], [0, 3, 3]]        # (z,y,x)
    assert n.tolist() == [2, 2, 1]
    assert v[0, :, 3].tolist() == [10, 12] and v[1, :, 3].tolist() == [11, 16] and v[2, :, 3].tolist() == [15, 0]


def test_hard_voxelize_refused_voxel_does_not_block_existing():
    vs, rng = [1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 2.0, 2.0, 1.0]
    pts = np.array([[0.5, 0.5, 0.5], [1.5, 0.5, 0.5], [1.5, 1.5, 0.5], [0.4, 0.4, 0.4]], dtype=np.float32)
    v, c, n = OC.hard_voxelize(pts, vs, rng, max_points=5, max_voxels=2)
    assert c.tolist() == [[0, 0, 0], [0, 0, 1]] and n.tolist() == [2, 1]


def test_hard_voxelize_against_an_independent_dictionary_restatement():
    """A second, independent statement of upstream's sequential algorithm (a dict from cell to voxel id, point by point) on random
    clouds that hit both caps and contain NaNs / out-of-range points; every output identical."""
    rng = np.random.default_rng(12)
    vs, cr = np.float32([0.5, 0.5, 4.0]), np.float32([-4.0, -3.0, -2.0, 4.0, 3.0, 2.0])
    grid = [int(round(float((cr[3 + a] - cr[a]) / vs[a]))) for a in range(3)]
    for trial, (n, mp, mv) in enumerate([(3000, 4, 150), (500, 10, 400), (2000, 1, 50)]):
        pts = rng.uniform(-4.6, 4.6, size=(n, 5)).astype(np.float32)
        pts[:, 2] = rng.uniform(-2.4, 2.4, size=n)
        pts[rng.integers(0, n, 5), 0] = np.nan
        ids, vox, coors, num = {}, [], [], []
        for p in pts:
            c = []
            for a in range(3):
                v = np.floor((np.float32(p[a]) - cr[a]) / vs[a])
                if not (v >= 0 and v < grid[a]):           # NaN fails both comparisons
                    c = None
                    break
                c.append(int(v))
            if c is None:
                continue
            key = (c[2], c[1], c[0])
            if key not in ids:
                if len(vox) >= mv:
                    continue
                ids[key] = len(vox)
                vox.append(np.zeros((mp, 5), np.float32)); coors.append(key); num.append(0)
            k = ids[key]
            if num[k] < mp:
                vox[k][num[k]] = p
                num[k] += 1
        v, c, m = OC.hard_voxelize(pts, vs, cr, max_points=mp, max_voxels=mv)
        assert c.tolist() == [list(k) for k in coors] and m.tolist() == num, trial
        assert np.array_equal(v, np.stack(vox), equal_nan=True), trial
        assert len(vox) == mv or trial == 1                 # trials 0 and 2 hit the voxel cap


def test_pillar_scatter_index():
    feats = np.arange(2 * 3, dtype=np.float32).reshape(2, 3) + 1
    coors = np.array([[0, 0, 1, 2], [1, 0, 0, 3]], dtype=np.int32)
    canvas = OC.pillar_scatter(feats, coors, batch=2, ny=2, nx=4)
    assert canvas.shape == (2, 3, 2, 4) and canvas.sum() == feats.sum()
    assert canvas[0, :, 1, 2].tolist() == [1, 2, 3] and canvas[1, :, 0, 3].tolist() == [4, 5, 6]


# ---------------------------------------------------------------------------------------------
# Rotated BEV IoU / NMS oracle (upstream mmdet3d v0.17.1 iou3d; parity unpinned, see nms_oracle.c)
# ---------------------------------------------------------------------------------------------
def _xyxyr(cx, cy, w, h, r):
    return [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, r]


def _clip_area(a, b):
    """Independent check: Sutherland-Hodgman clipping of polygon a by convex polygon b (float64)."""
    def corners(bx):
        cx, cy, w, h, r = (bx[0] + bx[2]) / 2, (bx[1] + bx[3]) / 2, bx[2] - bx[0], bx[3] - bx[1], bx[4]
        c, s = np.cos(r), np.sin(r)
        pts = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]])
        rot = np.array([[c, s], [-s, c]])              # the clockwise convention of the v0.17.1 kernel
        return pts @ rot.T + [cx, cy]
    cr = lambda u, v: u[0] * v[1] - u[1] * v[0]
    poly, clip = corners(np.asarray(a, np.float64)), corners(np.asarray(b, np.float64))
    if cr(clip[1] - clip[0], clip[2] - clip[1]) < 0:
        clip = clip[::-1]
    out = list(poly)
    for i in range(4):
        p0, p1 = clip[i], clip[(i + 1) % 4]
        inp, out = out, []
        if not inp:
            break
        side = lambda q: cr(p1 - p0, q - p0)
        for j in range(len(inp)):
            cur, prv = inp[j], inp[j - 1]
            if side(cur) >= 0:
                if side(prv) < 0:
                    out.append(prv + (cur - prv) * side(prv) / (side(prv) - side(cur)))
                out.append(cur)
            elif side(prv) >= 0:
                out.append(prv + (cur - prv) * side(prv) / (side(prv) - side(cur)))
    if len(out) < 3:
        return 0.0
    o = np.array(out)
    return 0.5 * abs(np.sum(o[:, 0] * np.roll(o[:, 1], -1) - np.roll(o[:, 0], -1) * o[:, 1]))


def test_iou_oracle_hand_computed_cases():
    a = np.array([_xyxyr(0, 0, 2, 2, 0)], np.float32)
    b = np.array([_xyxyr(1, 0, 2, 2, 0),            # half overlap: 2 / (4 + 4 - 2)
                  _xyxyr(0, 0, 2, 2, np.pi / 4),    # octagon of area 8(sqrt2 - 1)
                  _xyxyr(5, 5, 1, 1, 0.3),          # disjoint
                  _xyxyr(0, 0, 1, 1, 1.0),          # contained at any angle: 1 / 4
                  _xyxyr(0, 0, 2, 2, 0),            # identical
                  _xyxyr(0.5, 0.5, 4, 1, np.pi / 2)], np.float32)   # 1x4 bar turned upright: overlap 1 x 1.5 ... see below
    iou = OC.iou_bev_matrix(a, b)[0]
    oct_area = 8 * (2 ** 0.5 - 1)
    np.testing.assert_allclose(iou[:5], [1 / 3, oct_area / (8 - oct_area), 0.0, 0.25, 1.0], atol=1e-6)
    # bar: centre (0.5, 0.5), after the quarter turn it spans x in [0, 1], y in [-1.5, 2.5]; inside the 2x2
    # square that is 1 x 2 -> IoU 2 / (4 + 4 - 2)
    np.testing.assert_allclose(iou[5], 2 / 6, atol=1e-6)


def test_iou_oracle_against_independent_polygon_clipping():
    rng = np.random.default_rng(11)
    n = 60
    xy = rng.normal(0, 2.0, (n, 2))
    wl = np.stack([rng.uniform(0.5, 2.6, n), rng.uniform(0.5, 8.0, n)], 1)
    boxes = np.concatenate([xy - wl / 2, xy + wl / 2, rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)
    iou = OC.iou_bev_matrix(boxes, boxes)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    worst = 0.0
    for i in range(n):
        for j in range(n):
            ov = _clip_area(boxes[i], boxes[j])
            worst = max(worst, abs(ov / max(area[i] + area[j] - ov, 1e-8) - iou[i, j]))
    assert (iou > 0.05).sum() > n * 4
    assert worst < 2e-4, worst


def test_nms_oracle_greedy_semantics():
    boxes = np.array([_xyxyr(0, 0, 2, 4, 0.0), _xyxyr(0.2, 0, 2, 4, 0.05), _xyxyr(10, 0, 2, 4, 0.0),
                      _xyxyr(10, 0.1, 2, 4, 1.57), _xyxyr(0.1, 0.1, 2, 4, 0.0)], np.float32)
    scores = np.array([0.9, 0.8, 0.3, 0.95, 0.5], np.float32)
    # order: 3, 0, 1, 4, 2.  Box 0 suppresses 1 and 4; box 3 (upright vs lying) overlaps box 2 with
    # IoU = 4 / 12 > 0.2 -> suppressed at 0.2, kept at 0.4
    assert OC.nms_rotated(boxes, scores, 0.2).tolist() == [3, 0]
    assert OC.nms_rotated(boxes, scores, 0.4).tolist() == [3, 0, 2]
    assert OC.nms_rotated(boxes, scores, 0.2, pre_maxsize=2).tolist() == [3, 0]
    assert OC.nms_rotated(boxes, scores, 0.4, post_max_size=1).tolist() == [3]
    assert OC.nms_rotated(np.zeros((0, 5), np.float32), np.zeros(0, np.float32), 0.2).tolist() == []


# ---- deformable convolution v1 (mmcv 1.4.0, un-vendored): hand-computed cases; parity unpinned by the reference ----
def _dcn_inputs(rng, B=2, C=8, H=6, W=7, N=8, G=4):
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = rng.standard_normal((N, C // G, 3, 3)).astype(np.float32)
    return x, w


def test_dcn_oracle_zero_offsets_is_a_plain_grouped_convolution():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    x, w = _dcn_inputs(rng)
    for stride, pad, dil in ((1, 1, 1), (2, 1, 1), (1, 2, 2)):
        want = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, stride, pad, dil, 4).numpy()
        Ho, Wo = want.shape[2:]
        got = OC.deform_conv(x, np.zeros((2, 18, Ho, Wo), np.float32), w, stride, pad, dil, groups=4)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_dcn_oracle_offset_channel_order_is_dy_then_dx_per_tap():
    """One tap of a 3x3 kernel with weight 1, offset (dy, dx) = (+1, 0) on channel pair (2*t, 2*t + 1): the output is the
    input shifted UP by one more row for that tap (reads y + 1), not shifted along x."""
    x = np.arange(5 * 6, dtype=np.float32).reshape(1, 1, 5, 6)
    w = np.zeros((1, 1, 3, 3), np.float32)
    w[0, 0, 1, 1] = 1.0                                        # centre tap t = 4 reads x[y, x] when undeformed
    off = np.zeros((1, 18, 5, 6), np.float32)
    off[0, 2 * 4] = 1.0                                        # dy of the centre tap
    got = OC.deform_conv(x, off, w, 1, 1, 1)
    want = np.zeros_like(x)
    want[0, 0, :4] = x[0, 0, 1:]                               # row y + 1; the last row reads y = 5 = H: outside, 0
    assert np.array_equal(got, want)
    off[:] = 0
    off[0, 2 * 4 + 1] = -2.0                                   # dx of the centre tap
    got = OC.deform_conv(x, off, w, 1, 1, 1)
    want = np.zeros_like(x)
    want[0, 0, :, 2:] = x[0, 0, :, :-2]
    assert np.array_equal(got, want)


def test_dcn_oracle_border_rule_gt_minus_one_lt_size():
    """h in (-1, 0): only the in-range corner row contributes, with its bilinear weight; h <= -1 and h >= H: zero; h in
    (H-1, H): the row H-1 with weight H - h (mmcv: `h_im > -1 && ... && h_im < height`, corners outside contribute 0)."""
    x = np.full((1, 1, 4, 4), 2.0, np.float32)
    w = np.zeros((1, 1, 3, 3), np.float32)
    w[0, 0, 1, 1] = 1.0
    def at(dy, dx=0.0):
        off = np.zeros((1, 18, 4, 4), np.float32)
        off[0, 8], off[0, 9] = dy, dx
        return OC.deform_conv(x, off, w, 1, 1, 1)[0, 0]
    np.testing.assert_allclose(at(-0.25)[0], 2.0 * 0.75)       # h = -0.25: corner rows -1 (out) and 0 (weight 0.75)
    np.testing.assert_allclose(at(-1.0)[0], 0.0)
    np.testing.assert_allclose(at(-1.5)[0], 0.0)
    np.testing.assert_allclose(at(0.75)[3], 2.0 * 0.25)        # h = 3.75 on the last row: row 3 (weight 0.25), row 4 out
    np.testing.assert_allclose(at(1.0)[3], 0.0)                # h = 4 = H
    np.testing.assert_allclose(at(-0.5, -0.5)[0, 0], 2.0 * 0.25)   # one valid corner of four


def test_dcn_oracle_backward_is_the_gradient_of_its_forward():
    rng = np.random.default_rng(3)
    x, w = _dcn_inputs(rng, B=1, C=4, H=5, W=6, N=4, G=2)
    off = (rng.standard_normal((1, 18, 5, 6)) * 0.7).astype(np.float32)
    off += 0.013                                               # keep sampling positions off the integer grid
    g = rng.standard_normal((1, 4, 5, 6)).astype(np.float32)
    out, gx, go, gw = OC.deform_conv(x, off, w, 1, 1, 1, groups=2, grad_out=g)
    f = lambda x_, o_, w_: float((OC.deform_conv(x_, o_, w_, 1, 1, 1, groups=2).astype(np.float64) * g).sum())
    eps = 1e-2
    for arr, grad, name in ((x, gx, "x"), (off, go, "offset"), (w, gw, "weight")):
        for _ in range(12):
            idx = tuple(int(rng.integers(0, s)) for s in arr.shape)
            a, b = arr.copy(), arr.copy()
            a[idx] += eps; b[idx] -= eps
            args = {"x": (a, off, w), "offset": (x, a, w), "weight": (x, off, a)}[name], {"x": (b, off, w), "offset": (x, b, w), "weight": (x, off, b)}[name]
            num = (f(*args[0]) - f(*args[1])) / (2 * eps)
            assert abs(num - grad[idx]) <= 2e-2 * max(1.0, abs(num)), (name, idx, num, grad[idx])


@pytest.mark.parametrize("dg", [1, 2])
def test_dcn_oracle_agrees_with_the_two_torch_formulations_of_the_product(dg):
    """Three independent restatements of mmcv's deformable convolution — the C oracle (plain loops), the product's row-gather +
    GEMM formulation and its grid_sample formulation (omnihd_amd/mm/dcn.py, the CPU / reference paths of DeformConv2dPack) — on
    random fractional offsets that reach across the border (|offset| up to 2.5), groups = 4; 1e-5 of the largest output."""
    import torch
    from omnihd_amd.mm.dcn import DeformConv2dPack
    rng = np.random.default_rng(31 + dg)
    B, C, H, W, N = 2, 16, 7, 9, 16
    m = DeformConv2dPack(C, N, 3, stride=1, padding=1, dilation=1, groups=4, deform_groups=dg)
    with torch.no_grad():
        m.weight.copy_(torch.from_numpy(rng.normal(size=m.weight.shape).astype(np.float32)))
    x = rng.normal(size=(B, C, H, W)).astype(np.float32)
    off = rng.uniform(-2.5, 2.5, size=(B, dg * 18, H, W)).astype(np.float32)
    want = OC.deform_conv(x, off, m.weight.detach().numpy(), 1, 1, 1, groups=4, deform_groups=dg)
    with torch.no_grad():
        got = [m._sample_and_contract(torch.from_numpy(x), torch.from_numpy(off)).numpy()]
        if dg == 1:
            got.append(m._gather_and_gemm(torch.from_numpy(x), torch.from_numpy(off), torch.float32).numpy())
    for g in got:
        assert g.shape == want.shape and np.abs(g - want).max() <= 1e-5 * np.abs(want).max()


def test_pillar_feature_net_oracle_is_pinned_by_the_reference_outputs(golden):
    """oracle/pfn_oracle.py against what the reference's PillarFeatureNetV1 / RadarPillarFeatureNet classes produced on fixed
    inputs with fixed weights (BatchNorm in inference mode; tests/golden/make_golden.py keys g6_*)."""
    from oracle import pfn_oracle as P
    vsz, pcr = [0.25, 0.25, 8], [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]
    vox, npts, coors = golden["g6_voxels"], golden["g6_num_points"], golden["g6_coors"]
    x = P.decorate(vox, npts, coors, vsz, pcr)
    assert x.shape[-1] == 13
    w, b, rm, rv = golden["g6_pfn_bn"]
    out, _, _ = P.pfn_forward(x, golden["g6_pfn_linear_w"], w, b, rm, rv, eps=1e-3)
    np.testing.assert_allclose(out, golden["g6_pfn_out"], rtol=2e-6, atol=2e-6)
    sd = {k[len("g6_radar_sd__"):].replace("__", "."): golden[k] for k in golden.files if k.startswith("g6_radar_sd__")}
    xr = P.decorate(vox[:, :, :7], npts, coors, vsz, pcr, radar=True)
    assert xr.shape[-1] == 16
    wr = P.radar_weight(sd["pfn_layers.0.linear1.weight"], sd["pfn_layers.0.linear2.weight"], sd["pfn_layers.0.linear3.weight"])
    cat = lambda n: np.concatenate([sd[f"pfn_layers.0.norm{i}.{n}"] for i in (1, 2, 3)])
    out, _, _ = P.pfn_forward(xr, wr, cat("weight"), cat("bias"), cat("running_mean"), cat("running_var"), eps=1e-3)
    np.testing.assert_allclose(out, golden["g6_radar_out"], rtol=2e-6, atol=2e-6)
    # padded slots are rows of zeros that take part in the maximum (a pillar with one point still sees relu(shift))
    assert (npts < vox.shape[1]).any()
