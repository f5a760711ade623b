"""GPU parity tests of the radar-side ops: hard voxelisation (bit-exact voxel assignment) and
pillar scatter, against the sequential CPU oracle (upstream mmdet3d semantics; see
oracle/voxelize_oracle.c for the pinning status)."""
import numpy as np
import pytest
import torch

from oracle import cpu as OC
from tests.helpers import t

pytestmark = pytest.mark.gpu
VS = [0.25, 0.25, 8]
RNG6 = [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]


def radar_cloud(rng, n, f=8, spread=1.0):
    pts = np.empty((n, f), dtype=np.float32)
    pts[:, 0] = rng.uniform(-60 * spread, 60 * spread, n)
    pts[:, 1] = rng.uniform(-40 * spread, 40 * spread, n)
    pts[:, 2] = rng.uniform(-3, 5, n)
    pts[:, 3:] = rng.standard_normal((n, f - 3))
    return pts


def check(cuda, pts, vs, rng6, max_points, max_voxels):
    from omnihd_amd import ops
    want = OC.hard_voxelize(pts, vs, rng6, max_points, max_voxels)
    got = ops.hard_voxelize(t(pts, cuda), vs, rng6, max_points, max_voxels)
    assert got[1].dtype == torch.int32 and got[2].dtype == torch.int32
    assert np.array_equal(got[1].cpu().numpy(), want[1])      # coors (z,y,x), first-occurrence order
    assert np.array_equal(got[2].cpu().numpy(), want[2])      # points per voxel
    assert np.array_equal(got[0].cpu().numpy(), want[0])      # point rows, zero padded
    return want


@pytest.mark.parametrize("n,f", [(12000, 8), (20000, 7), (1, 8), (257, 5)])
def test_voxelize_random_clouds(cuda, n, f):
    rng = np.random.default_rng(n + f)
    w = check(cuda, radar_cloud(rng, n, f, spread=1.05), VS, RNG6, 10, 30000)
    assert len(w[1]) > 0


def test_voxelize_dense_clusters_hit_max_points(cuda):
    rng = np.random.default_rng(1)
    pts = radar_cloud(rng, 5000)
    pts[:3000, 0] = rng.uniform(0, 1.0, 3000)       # 3000 points into 4x4 cells
    pts[:3000, 1] = rng.uniform(0, 1.0, 3000)
    w = check(cuda, pts, VS, RNG6, 10, 30000)
    assert w[2].max() == 10


def test_voxelize_max_voxels_cap(cuda):
    rng = np.random.default_rng(2)
    w = check(cuda, radar_cloud(rng, 20000), VS, RNG6, 10, 500)
    assert len(w[1]) == 500


def test_voxelize_edges_nan_and_all_outside(cuda):
    from omnihd_amd import ops
    pts = np.array([[-60.0, -40.0, -3.0, 1], [60.0, 0, 0, 2], [59.99999, 39.99999, 4.99999, 3],
                    [np.nan, 0, 0, 4], [0, np.inf, 0, 5], [-60.00001, 0, 0, 6], [0, 0, 5.0, 7],
                    [-59.75, -39.75, 0, 8], [-59.76, -39.76, 1, 9]], dtype=np.float32)
    check(cuda, pts, VS, RNG6, 10, 100)
    out = ops.hard_voxelize(t(np.full((50, 4), 1e6, np.float32), cuda), VS, RNG6, 10, 100)
    assert out[0].shape == (0, 10, 4) and out[1].shape == (0, 3) and out[2].shape == (0,)
    out = ops.hard_voxelize(torch.empty(0, 4, device=cuda), VS, RNG6, 10, 100)
    assert out[0].shape[0] == 0


def test_three_launch_voxelisation_equals_the_sort_path_and_the_oracle(cuda, monkeypatch):
    """Round 5: the per-cell atomicMin / list / one-workgroup scan path (three launches, persistent idle cell state) against
    the stable-sort path and the sequential oracle: one crowded cell (5 000 points: the kept points are the FIRST ten by
    index, whatever order the atomics arrived in), both caps binding, NaN / out-of-range points, an empty cloud, and repeated
    calls on the same persistent state."""
    from omnihd_amd import ops
    rng = np.random.default_rng(17)
    clouds = []
    crowd = np.tile(np.array([[1.1, 2.2, 0.5]], np.float32), (5000, 1)) + rng.uniform(0, 0.2, (5000, 3)).astype(np.float32)
    clouds.append(np.concatenate([crowd, rng.standard_normal((5000, 4)).astype(np.float32)], 1))
    dense = np.concatenate([rng.uniform(-60, 60, (20000, 1)), rng.uniform(-40, 40, (20000, 1)), rng.uniform(-3, 5, (20000, 1)),
                            rng.standard_normal((20000, 4))], 1).astype(np.float32)
    dense[::97, 0] = np.nan; dense[::89, 1] = 1e9
    clouds.append(dense)
    small = dense[:3000].copy(); small[:, :2] *= 0.05                     # many points per cell: the 10-point cap binds
    clouds.append(small)
    for pts in clouds:
        for max_voxels in (30000, 700):                                   # 700: the voxel cap binds
            want = OC.hard_voxelize(pts, VS, RNG6, 10, max_voxels)
            monkeypatch.setenv("OMNIHD_VOXELIZE_GRID", "1")
            got = ops.hard_voxelize(t(pts, cuda), VS, RNG6, 10, max_voxels)
            monkeypatch.setenv("OMNIHD_VOXELIZE_GRID", "0")
            srt = ops.hard_voxelize(t(pts, cuda), VS, RNG6, 10, max_voxels)
            for g, s_, w in zip(got, srt, want):
                assert g.shape == w.shape and np.array_equal(g.cpu().numpy(), w) and torch.equal(g, s_)
    monkeypatch.setenv("OMNIHD_VOXELIZE_GRID", "1")
    out = ops.hard_voxelize(torch.empty(0, 7, device=cuda), VS, RNG6, 10, 100)
    assert out[0].shape[0] == 0
    state = [v[0] for k, v in ops._VOXEL_STATE.items() if v[0] is not None]
    assert state, "the grid path did not run"
    for buf in state:                                                     # idle again after every call
        words = buf.view(torch.int32)
        half = words.numel() // 2
        assert bool((words[:half] == 0x7fffffff).all()) and bool((words[half:] == -1).all())


@pytest.mark.parametrize("channels_last", [False, True])
def test_pillar_scatter_and_backward(cuda, channels_last):
    from omnihd_amd import ops
    rng = np.random.default_rng(4)
    B, ny, nx, C = 2, 320, 480, 64
    coors = []
    for b in range(B):
        cells = rng.permutation(ny * nx)[:9000]
        coors.append(np.stack([np.full(9000, b), np.zeros(9000, int), cells // nx, cells % nx], 1))
    coors = np.concatenate(coors).astype(np.int32)
    feats = rng.standard_normal((len(coors), C), dtype=np.float32)
    want = OC.pillar_scatter(feats, coors, B, ny, nx)
    f = t(feats, cuda).requires_grad_()
    canvas = ops.pillar_scatter(f, t(coors, cuda), B, ny, nx, channels_last=channels_last)
    assert canvas.shape == (B, C, ny, nx)
    assert np.array_equal(canvas.detach().cpu().numpy(), want)
    w = torch.randn(B, C, ny, nx, device=cuda)
    (canvas * w).sum().backward()
    c = torch.from_numpy(coors).long().to(cuda)
    assert torch.equal(f.grad, w[c[:, 0], :, c[:, 2], c[:, 3]])


@pytest.mark.parametrize("C", [8, 48, 96, 256, 320])
def test_channels_last_scatter_with_channel_counts_whose_cells_straddle_wavefronts(cuda, C):
    """the float4 canvas kernel resets the caller-kept cell map in the kernel only when the C/4 lanes of a cell
    sit in one wavefront (C/4 a power of two <= 64); C = 48, 96, 320 must take the separate reset.  Twice on the same map: the
    second call sees whatever the first left behind."""
    from omnihd_amd import ops
    rng = np.random.default_rng(C)
    B, ny, nx = 1, 96, 160
    for rep in range(2):
        cells = rng.permutation(ny * nx)[:6000]
        coors = np.stack([np.zeros(6000, int), np.zeros(6000, int), cells // nx, cells % nx], 1).astype(np.int32)
        feats = rng.standard_normal((6000, C), dtype=np.float32)
        canvas = ops.pillar_scatter(t(feats, cuda), t(coors, cuda), B, ny, nx, channels_last=True)
        assert np.array_equal(canvas.cpu().numpy(), OC.pillar_scatter(feats, coors, B, ny, nx)), (C, rep)


def test_pillar_scatter_odd_plane_and_empty(cuda):
    from omnihd_amd import ops
    feats = torch.arange(6, dtype=torch.float32, device=cuda).view(2, 3) + 1
    coors = torch.tensor([[0, 0, 1, 2], [1, 0, 0, 0]], dtype=torch.int32, device=cuda)
    canvas = ops.pillar_scatter(feats, coors, 2, 3, 3)        # plane of 9: scalar path
    want = OC.pillar_scatter(feats.cpu().numpy(), coors.cpu().numpy(), 2, 3, 3)
    assert np.array_equal(canvas.cpu().numpy(), want)
    empty = ops.pillar_scatter(torch.empty(0, 3, device=cuda), torch.empty(0, 4, dtype=torch.int32, device=cuda), 1, 4, 4)
    assert empty.abs().sum() == 0


def test_radar_sweep_merge_on_the_device_matches_the_host_loader(cuda, tmp_path):
    """csrc/radar_merge.hip against the host loader (itself bit-identical to the reference's, tests/test_data_cpu.py) on
    the golden sweeps: positions bit-exact (they decide the voxel), velocity columns to float32-trig accuracy."""
    import os
    from projects.mmdet3d_plugin.datasets.pipelines.loading import (RADAR_ID, LoadRadarPointsMultiSweeps, merge_radar_sweeps,
                                                                    merge_radar_sweeps_device)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = np.load(os.path.join(root, "tests", "golden", "data_golden.npz"))
    names, radars, off = list(RADAR_ID), {}, 0
    for row in gold["radar_meta"]:
        ri, si, n, ts = int(row[0]), int(row[1]), int(row[2]), int(row[3])
        path = os.path.join(str(tmp_path), f"r{ri}_{si}.bin")
        gold["radar_raw"][off:off + n].astype(np.float32).tofile(path)
        off += n
        radars.setdefault(names[ri], []).append(dict(data_path=path, timestamp=ts, ego_velocity=row[4:7].tolist(),
                                                     sensor2ego_rotation=row[7:11].tolist(),
                                                     sensor2lidar_rotation=row[11:20].reshape(3, 3),
                                                     sensor2lidar_translation=row[20:23]))
    rng6 = [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]
    want = merge_radar_sweeps(radars, LoadRadarPointsMultiSweeps._load_points, 3, 8).astype(np.float32)
    got, mask = merge_radar_sweeps_device(radars, LoadRadarPointsMultiSweeps._load_points, cuda, 3, 8, rng6)
    got = got.cpu().numpy()
    assert got.shape == want.shape and len(got) > 1000
    assert np.array_equal(got[:, :3], want[:, :3])                         # positions: bit for bit
    assert np.array_equal(got[:, [5, 6, 7, 9]], want[:, [5, 6, 7, 9]])       # power, snr, dt, radar id: copies
    np.testing.assert_allclose(got[:, [3, 4, 8]], want[:, [3, 4, 8]], rtol=2e-5, atol=2e-5)
    wm = ((want[:, 0] > -60) & (want[:, 1] > -40) & (want[:, 2] > -3) & (want[:, 0] < 60) & (want[:, 1] < 40) & (want[:, 2] < 5))
    assert np.array_equal(mask.cpu().numpy(), wm) and 0 < wm.sum() < len(wm)
    host = LoadRadarPointsMultiSweeps(load_dim=8, sweeps_num=3, use_dim=list(range(8)), pc_range=rng6)({"radars": radars})["points"]
    dev = LoadRadarPointsMultiSweeps(load_dim=8, sweeps_num=3, use_dim=list(range(8)), pc_range=rng6, device=cuda)({"radars": radars})["points"]
    assert dev.tensor.is_cuda and dev.tensor.shape == host.tensor.shape
    assert torch.equal(dev.tensor[:, :3].cpu(), host.tensor[:, :3])
    torch.testing.assert_close(dev.tensor.cpu(), host.tensor, rtol=2e-5, atol=2e-5)
    # and straight into the voxeliser: same voxel assignment as from the host-loaded points
    from omnihd_amd import ops
    a = ops.hard_voxelize(dev.tensor.contiguous(), VS, RNG6, 10, 30000)
    b = ops.hard_voxelize(host.tensor.to(cuda).contiguous(), VS, RNG6, 10, 30000)
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
