"""oracle/lss_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT.

numpy restatement of the Lift-Splat geometry and rank-table logic of the reference
(``projects/mmdet3d_plugin/bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py``; file:line
cited per function).  Pinned by ``tests/test_oracle.py`` against golden vectors captured from the
reference Python itself (``tests/golden/make_golden.py``).
"""
import numpy as np

F32 = np.float32


def gen_dx_bx(xbound, ybound, zbound):
    """ref: cam_stream_lss_bevpoolv2_depthnet.py:80-85.

    ``torch.Tensor([python floats])`` rounds each python double to fp32 once; ``LongTensor`` of a
    python float truncates toward zero.
    """
    rows = [xbound, ybound, zbound]
    dx = np.array([row[2] for row in rows], dtype=np.float64).astype(F32)
    bx = np.array([row[0] + row[2] / 2.0 for row in rows], dtype=np.float64).astype(F32)
    nx = np.array([int((row[1] - row[0]) / row[2]) for row in rows], dtype=np.int64)
    return dx, bx, nx


def torch_linspace_f32(start, end, steps):
    """ATen's CPU linspace for float32: the step is computed in fp32, the first half counts up
    from ``start`` and the second half counts back from ``end``; each element is ONE fused
    multiply-add (a single rounding), which is emulated here by doing the product and the sum in
    float64 (exact for fp32 operands) and rounding once.  Verified bit-exact against the
    reference's frustum for every resolution in tests/golden (see tests/test_oracle.py)."""
    start, end = F32(start), F32(end)
    if steps == 1:
        return np.array([start], dtype=F32)
    step = np.float64(F32((end - start) / F32(steps - 1)))
    idx = np.arange(steps, dtype=np.float64)
    half = steps // 2
    out = np.empty(steps, dtype=F32)
    out[:half] = (np.float64(start) + step * idx[:half]).astype(F32)
    out[half:] = (np.float64(end) - step * (steps - idx[half:] - 1)).astype(F32)
    return out


def frustum_axes(final_dim, downsample, dbound):
    """ref: create_frustum, cam_stream_lss_bevpoolv2_depthnet.py:222-233.  The (D,fH,fW,3) frustum
    is the broadcast of three vectors; return them (xs over fW, ys over fH, ds over D)."""
    ogfH, ogfW = final_dim
    fH, fW = ogfH // downsample, ogfW // downsample
    ds = np.arange(dbound[0], dbound[1], dbound[2], dtype=np.float64).astype(F32)
    # torch.arange(*dbound, dtype=float): size = ceil((end-start)/step), value = start + i*step
    n = int(np.ceil((dbound[1] - dbound[0]) / dbound[2]))
    ds = (np.float64(dbound[0]) + np.arange(n, dtype=np.float64) * np.float64(dbound[2])).astype(F32)
    xs = torch_linspace_f32(0, ogfW - 1, fW)
    ys = torch_linspace_f32(0, ogfH - 1, fH)
    return xs, ys, ds


def create_frustum(final_dim, downsample, dbound):
    xs, ys, ds = frustum_axes(final_dim, downsample, dbound)
    D, fH, fW = len(ds), len(ys), len(xs)
    fr = np.empty((D, fH, fW, 3), dtype=F32)
    fr[..., 0] = xs[None, None, :]
    fr[..., 1] = ys[None, :, None]
    fr[..., 2] = ds[:, None, None]
    return fr


def get_geometry(frustum, rots, trans):
    """ref: get_geometry, cam_stream_lss_bevpoolv2_depthnet.py:235-264 with post_*/extra_* = None
    (the call at bevf_faster_rcnn_bevdepth.py:133 passes none of them).

    p = (u*d, v*d, d); geom = rots @ p + trans, all fp32.  The 3-term dot product is evaluated as
    ((r0*p0 + r1*p1) + r2*p2) without fusing; backends may round differently in the last ulp, so
    consumers compare geometry with a tolerance and rank tables on a SHARED geometry tensor.
    """
    rots = np.asarray(rots, dtype=F32)
    trans = np.asarray(trans, dtype=F32)
    B, N = trans.shape[:2]
    px = frustum[..., 0] * frustum[..., 2]
    py = frustum[..., 1] * frustum[..., 2]
    pz = frustum[..., 2]
    out = np.empty((B, N) + frustum.shape, dtype=F32)
    for b in range(B):
        for n in range(N):
            R = rots[b, n]
            for a in range(3):
                acc = (R[a, 0] * px + R[a, 1] * py).astype(F32)
                acc = (acc + R[a, 2] * pz).astype(F32)
                out[b, n, ..., a] = acc + trans[b, n, a]
    return out


def voxel_pooling_prepare_v2(coor, dx, bx, nx):
    """ref: voxel_pooling_prepare_v2, cam_stream_lss_bevpoolv2_depthnet.py:302-362.

    Returns (ranks_bev, ranks_depth, ranks_feat, interval_starts, interval_lengths) as int32, in
    the CANONICAL order (stable sort; the reference's argsort is unstable by contract, defect D6,
    but torch-CPU is stable on this data), or five ``None`` when no point survives (the reference's
    guards at :338-339 / :354-355; defect D4 is documented, not reproduced).
    """
    coor = np.asarray(coor, dtype=F32)
    B, N, D, H, W, _ = coor.shape
    num_points = B * N * D * H * W
    ranks_depth = np.arange(num_points, dtype=np.int32)                      # :316-317
    ranks_feat = np.arange(num_points // D, dtype=np.int32).reshape(B, N, 1, H, W)
    ranks_feat = np.broadcast_to(ranks_feat, (B, N, D, H, W)).reshape(-1)    # :318-322
    off = (bx - dx / F32(2.0)).astype(F32)                                   # fp32 tensor math
    t = ((coor - off) / dx).astype(F32)                                      # :324
    with np.errstate(invalid="ignore"):
        c = t.astype(np.int64).reshape(num_points, 3)                        # .long(): toward zero (D3)
    batch_idx = np.repeat(np.arange(B, dtype=np.int64), num_points // B)     # :326-328
    kept = ((c[:, 0] >= 0) & (c[:, 0] < nx[0]) & (c[:, 1] >= 0) & (c[:, 1] < nx[1]) &
            (c[:, 2] >= 0) & (c[:, 2] < nx[2]))                              # :331-333
    if kept.sum() == 0:
        return None, None, None, None, None
    c, ranks_depth, ranks_feat, batch_idx = c[kept], ranks_depth[kept], ranks_feat[kept], batch_idx[kept]
    ranks_bev = batch_idx * (nx[2] * nx[1] * nx[0])                          # :339-342
    ranks_bev = ranks_bev + c[:, 2] * (nx[1] * nx[0])
    ranks_bev = ranks_bev + c[:, 1] * nx[0] + c[:, 0]
    order = np.argsort(ranks_bev, kind="stable")                             # :343
    ranks_bev, ranks_depth, ranks_feat = ranks_bev[order], ranks_depth[order], ranks_feat[order]
    starts, lengths = run_length(ranks_bev)                                  # :346-355
    return (ranks_bev.astype(np.int32), ranks_depth.astype(np.int32), ranks_feat.astype(np.int32),
            starts, lengths)


def run_length(sorted_keys):
    """interval_starts / interval_lengths of equal-key runs (ref :346-355, bev_pool.py:50-57)."""
    n = sorted_keys.shape[0]
    kept = np.ones(n, dtype=bool)
    kept[1:] = sorted_keys[1:] != sorted_keys[:-1]
    starts = np.nonzero(kept)[0].astype(np.int32)
    lengths = np.zeros_like(starts)
    lengths[:-1] = starts[1:] - starts[:-1]
    lengths[-1] = n - starts[-1]
    return starts, lengths


def backward_tables(ranks_bev, ranks_depth, ranks_feat):
    """ref: QuickCumsumCuda.backward, ops/bev_pool_v2/bev_pool.py:47-57 — re-sort by ranks_feat
    (stable = canonical) and rebuild the intervals."""
    order = np.argsort(ranks_feat, kind="stable")
    rf, rd, rb = ranks_feat[order], ranks_depth[order], ranks_bev[order]
    starts, lengths = run_length(rf)
    return rb, rd, rf, starts, lengths


def synthetic_rig(H, W, fx, yaws_deg=(0, 60, -60, 180, 120, -120), radius=1.0, height=1.5):
    """The 6-camera synthetic rig of SURVEY.md Appendix C: float64 lidar2img 4x4 per camera."""
    mats = []
    for yaw_deg in yaws_deg:
        yaw = np.radians(yaw_deg)
        R_c2l = np.array([[np.sin(yaw), 0, np.cos(yaw)],
                          [-np.cos(yaw), 0, np.sin(yaw)],
                          [0, -1, 0]], dtype=np.float64)
        t_c2l = np.array([radius * np.cos(yaw), radius * np.sin(yaw), height])
        R_l2c = R_c2l.T
        t_l2c = -R_l2c @ t_c2l
        K4 = np.eye(4)
        K4[0, 0] = K4[1, 1] = fx
        K4[0, 2] = W / 2
        K4[1, 2] = H / 2
        E = np.eye(4)
        E[:3, :3] = R_l2c
        E[:3, 3] = t_l2c
        mats.append(K4 @ E)
    return np.stack(mats)
