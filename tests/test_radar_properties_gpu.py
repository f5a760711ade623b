"""Size-independent properties of the radar-side operators at frame-sized clouds (120 000 points: six radar sweeps of 20 000),
where the sequential oracle of hard voxelisation (O(points x voxels) search, upstream mmdet3d voxelization semantics) is too slow
to be the checker: what ANY correct hard voxelisation / pillar scatter must satisfy.
  * voxelise: distinct voxel coordinates inside the grid; 1 <= points per voxel <= max_points; with caps that do not bind every
    finite in-range point is stored exactly once, in its own cell, rows past a voxel's count are zero; voxels appear in the order
    of their first point; the operator is idempotent on its own output (a cloud made of the stored points gives the same voxels);
  * scatter / gather: gather(scatter(x)) = x for distinct cells (round trip), scatter is linear, cells nobody names stay zero."""
import numpy as np
import pytest
import torch

from tests.helpers import t
from tests.test_radar_gpu import RNG6, VS, radar_cloud

pytestmark = pytest.mark.gpu
NX, NY = 480, 320


def _cells(p):
    p = np.where(np.isfinite(p), p, np.float32(1e9))                  # invalid returns: far outside the grid
    cx = np.floor((p[:, 0] - RNG6[0]) / VS[0]).astype(np.int64)
    cy = np.floor((p[:, 1] - RNG6[1]) / VS[1]).astype(np.int64)
    cz = np.floor((p[:, 2] - RNG6[2]) / VS[2]).astype(np.int64)
    ok = (cx >= 0) & (cx < NX) & (cy >= 0) & (cy < NY) & (cz == 0)
    return cx, cy, ok


@pytest.mark.parametrize("max_points,max_voxels,binding", [(64, 160000, False), (10, 30000, True)])
def test_hard_voxelise_invariants_on_a_frame_sized_cloud(cuda, max_points, max_voxels, binding):
    from omnihd_amd import ops
    rng = np.random.default_rng(7)
    pts = radar_cloud(rng, 120000, 7, spread=1.03)
    pts[rng.integers(0, len(pts), 40), 1] = np.nan                   # a few invalid returns
    pts[:, 3] = np.arange(len(pts), dtype=np.float32)                # channel 3 = point id (exact in fp32 up to 2^24)
    vox, coors, num = (a.cpu().numpy() for a in ops.hard_voxelize(t(pts, cuda), VS, RNG6, max_points, max_voxels))
    M = len(coors)
    assert 0 < M <= max_voxels and num.min() >= 1 and num.max() <= max_points
    assert (coors[:, 0] == 0).all() and (coors[:, 1] >= 0).all() and (coors[:, 1] < NY).all() and (coors[:, 2] < NX).all()
    assert len(np.unique(coors[:, 1].astype(np.int64) * NX + coors[:, 2])) == M          # one voxel per cell
    slot = np.arange(max_points)[None, :] < num[:, None]
    assert not vox[~slot].any()                                                          # rows past the count are zero padding
    ids = vox[..., 3][slot].astype(np.int64)
    assert len(np.unique(ids)) == len(ids)                                               # no point stored twice
    cx, cy, ok = _cells(pts)
    vi = np.repeat(np.arange(M), num)
    assert ok[ids].all() and (cx[ids] == coors[vi, 2]).all() and (cy[ids] == coors[vi, 1]).all()      # every stored point lies in its voxel's cell
    assert np.array_equal(vox[slot], pts[ids])                                           # stored rows are the input rows, bit for bit
    first = np.full(M, len(pts), np.int64)
    np.minimum.at(first, vi, ids)
    assert (np.diff(first) > 0).all()                                                    # voxels in the order of their first point
    assert (np.diff(ids.reshape(-1))[np.diff(vi) == 0] > 0).all()                        # inside a voxel: input order
    if not binding:
        assert len(ids) == int(ok.sum())                                                 # nothing dropped: every valid point exactly once
    # idempotence: the stored points, fed back in storage order, reproduce the same voxels
    again = [a.cpu().numpy() for a in ops.hard_voxelize(t(np.ascontiguousarray(vox[slot]), cuda), VS, RNG6, max_points, max_voxels)]
    assert np.array_equal(again[1], coors) and np.array_equal(again[2], num) and np.array_equal(again[0], vox)


@pytest.mark.parametrize("channels_last", [False, True])
def test_pillar_scatter_round_trip_and_linearity(cuda, channels_last):
    from omnihd_amd import ops
    rng = np.random.default_rng(9)
    B, C, M = 2, 64, 30000
    coors = []
    for b in range(B):
        cells = rng.permutation(NY * NX)[:M]
        coors.append(np.stack([np.full(M, b), np.zeros(M, int), cells // NX, cells % NX], 1))
    coors = t(np.concatenate(coors).astype(np.int32), cuda)
    f1, f2 = torch.randn(B * M, C, device=cuda), torch.randn(B * M, C, device=cuda)
    c = coors.long()
    s = lambda f: ops.pillar_scatter(f, coors, B, NY, NX, channels_last=channels_last)
    canvas = s(f1)
    assert torch.equal(canvas[c[:, 0], :, c[:, 2], c[:, 3]], f1)                         # round trip: gather(scatter(x)) == x
    named = torch.zeros(B, NY, NX, dtype=torch.bool, device=cuda)
    named[c[:, 0], c[:, 2], c[:, 3]] = True
    assert not bool(canvas.permute(0, 2, 3, 1)[~named].any())                            # cells nobody names stay zero
    assert int(named.sum()) == B * M
    assert torch.equal(s(0.5 * f1 + f2), 0.5 * canvas + s(f2))                           # linear (each cell is one row: exact)
    # the backward is the gather: the adjoint of the scatter
    x = f1.clone().requires_grad_()
    w = torch.randn(B, C, NY, NX, device=cuda)
    (s(x) * w).sum().backward()
    assert torch.equal(x.grad, w[c[:, 0], :, c[:, 2], c[:, 3]])
