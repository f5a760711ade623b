"""Fused pillar feature net (csrc/pillar_pfn.hip: decorate + Linear + BatchNorm + ReLU + max in 4 launches forward, 3 backward)
against (a) what the REFERENCE classes produced on fixed inputs (golden g6_*, BatchNorm in inference mode), (b) the float64
numpy oracle oracle/pfn_oracle.py (itself pinned by those goldens) in training mode, (c) the torch formulation of the module
mirror (OMNIHD_PFN_FUSED=0, float64 on the CPU) for the gradients and the running statistics, and (d) across two ranks the
reference's naiveSyncBN semantics (mean of rank means)."""
import os

import numpy as np
import pytest
import torch

from oracle import pfn_oracle as PO
from tests.helpers import t

pytestmark = pytest.mark.gpu
VSZ, PCR = [0.25, 0.25, 8], [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]
NCFG = dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01)


def _nets():
    from projects.mmdet3d_plugin.rcfusion.voxel_encoders import PillarFeatureNetV1, RadarPillarFeatureNet
    return PillarFeatureNetV1, RadarPillarFeatureNet


def _pillars(rng, n_points, f):
    """Real pillars: a radar cloud through the oracle's sequential hard voxeliser (pillars with 1..10 points, padded slots)."""
    from oracle import cpu as OC
    pts = np.empty((n_points, f), dtype=np.float32)
    pts[:, 0] = rng.uniform(-60, 60, n_points); pts[:, 1] = rng.uniform(-40, 40, n_points); pts[:, 2] = rng.uniform(-3, 5, n_points)
    pts[: n_points // 3, :2] = pts[n_points // 3: 2 * (n_points // 3), :2] + rng.normal(0, 0.05, (n_points // 3, 2)).astype(np.float32)
    pts[:, 3:5] = rng.normal(0, 5, (n_points, 2)); pts[:, 5] = rng.uniform(0, 60, n_points); pts[:, 6] = rng.uniform(0, 40, n_points)
    if f > 7:
        pts[:, 7] = rng.integers(0, 3, n_points) * 0.1
    vox, coors, num = OC.hard_voxelize(pts, VSZ, PCR, 10, 30000)
    coors = np.concatenate([np.zeros((len(coors), 1), np.int32), coors], 1)
    return vox, num, coors


def test_inference_mode_matches_the_reference_outputs(cuda, golden, monkeypatch):
    """BatchNorm with running statistics: the fused kernels against the outputs of the reference's own classes (1e-5 as the
    CPU mirror is held to; measured ~1e-6: the reference's Linear is a BLAS call, ours an fma chain in channel order)."""
    PillarFeatureNetV1, RadarPillarFeatureNet = _nets()
    net = PillarFeatureNetV1(in_channels=8, feat_channels=[64], with_distance=False, voxel_size=VSZ, point_cloud_range=PCR, norm_cfg=NCFG)
    net.pfn_layers[0].linear.weight.data = torch.from_numpy(golden["g6_pfn_linear_w"])
    bn = net.pfn_layers[0].norm
    w, b, rm, rv = [torch.from_numpy(x) for x in golden["g6_pfn_bn"]]
    bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var = w, b, rm, rv
    net = net.to(cuda).eval()
    vox, npts, coors = (t(golden[k], cuda) for k in ("g6_voxels", "g6_num_points", "g6_coors"))
    calls = []
    from omnihd_amd import ops
    real = ops.pfn_fused
    monkeypatch.setattr(ops, "pfn_fused", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        y = net(vox, npts, coors)
    assert calls, "the fused kernels did not run"
    np.testing.assert_allclose(y.cpu().numpy(), golden["g6_pfn_out"], rtol=1e-5, atol=1e-5)
    assert torch.equal(bn.running_mean.cpu(), rm) and torch.equal(bn.running_var.cpu(), rv)          # untouched in eval mode
    rnet = RadarPillarFeatureNet(in_channels=7, feat_channels=[64], with_distance=False, voxel_size=VSZ, point_cloud_range=PCR, norm_cfg=NCFG)
    sd = {k[len("g6_radar_sd__"):].replace("__", "."): torch.from_numpy(golden[k]) for k in golden.files if k.startswith("g6_radar_sd__")}
    rnet.load_state_dict(sd, strict=False)
    rnet = rnet.to(cuda).eval()
    n0 = len(calls)
    with torch.no_grad():
        y7 = rnet(vox[:, :, :7].contiguous(), npts, coors)
    assert len(calls) == n0 + 1
    np.testing.assert_allclose(y7.cpu().numpy(), golden["g6_radar_out"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("variant,f,legacy,distance", [("pfn", 8, True, False), ("pfn", 7, False, True), ("radar", 7, True, False),
                                                        ("pfn", 4, True, False)])
def test_training_mode_forward_backward_and_running_statistics(cuda, variant, f, legacy, distance, monkeypatch):
    """Batch statistics from the moments of the decorated points, ~18 k real pillars: output vs the float64 oracle (1e-5 of the
    largest value), gradients of Linear / BatchNorm weights and the running statistics vs the torch formulation of the module
    in float64 on the CPU (1e-4 relative L2; fp32 sums over 180 k rows), run-to-run identical."""
    PillarFeatureNetV1, RadarPillarFeatureNet = _nets()
    rng = np.random.default_rng(f * 10 + legacy + 2 * distance)
    vox, num, coors = _pillars(rng, 19000, max(f, 7))
    vox = np.ascontiguousarray(vox[:, :, :f])
    cls = RadarPillarFeatureNet if variant == "radar" else PillarFeatureNetV1
    torch.manual_seed(f)
    net = cls(in_channels=f, feat_channels=[64], with_distance=distance, voxel_size=VSZ, point_cloud_range=PCR, norm_cfg=NCFG, legacy=legacy)
    for n, p_ in net.named_parameters():                 # non-trivial affine parameters
        if "norm" in n:
            p_.data = torch.rand_like(p_) + 0.5 if n.endswith("weight") else torch.randn_like(p_) * 0.2
    import copy
    ref = copy.deepcopy(net).train()
    net = net.to(cuda).train()
    gy = torch.randn(len(vox), 64, generator=torch.Generator().manual_seed(1))
    # --- fused path on the GPU
    y = net(t(vox, cuda), t(num, cuda), t(coors, cuda))
    assert y.shape == (len(vox), 64) and y.dtype == torch.float32
    y.backward(gy.to(cuda))
    got = {n: p_.grad.detach().cpu().double() for n, p_ in net.named_parameters()}
    # --- oracle (forward, statistics)
    x = PO.decorate(vox, num, coors, VSZ, PCR, distance=distance, legacy=legacy, radar=variant == "radar")
    layer = ref.pfn_layers[0]
    if variant == "radar":
        W = PO.radar_weight(*[getattr(layer, f"linear{i}").weight.detach().numpy() for i in (1, 2, 3)], k=x.shape[-1])
        gam = np.concatenate([getattr(layer, f"norm{i}").weight.detach().numpy() for i in (1, 2, 3)])
        bet = np.concatenate([getattr(layer, f"norm{i}").bias.detach().numpy() for i in (1, 2, 3)])
    else:
        W, gam, bet = layer.linear.weight.detach().numpy(), layer.norm.weight.detach().numpy(), layer.norm.bias.detach().numpy()
    want, mean, var = PO.pfn_forward(x, W, gam, bet, training=True)
    scale = np.abs(want).max()
    assert np.abs(y.detach().cpu().numpy() - want).max() <= 1e-5 * scale
    # --- the reference's formulation (Linear -> BatchNorm1d with batch statistics -> ReLU -> max, utils.py:160-168) in float64
    # on the decorated points: gradients by autograd, running statistics by torch's update rule (unbiased variance)
    Wt, gt, bt = (torch.from_numpy(np.asarray(a_, dtype=np.float64)).requires_grad_() for a_ in (W, gam, bet))
    z = (torch.from_numpy(x) @ Wt.t()).reshape(-1, 64)
    mu, va = z.mean(0), z.var(0, unbiased=False)
    out = torch.relu((z - mu) * torch.rsqrt(va + 1e-3) * gt + bt).reshape(len(vox), -1, 64).max(1)[0]
    np.testing.assert_allclose(out.detach().numpy(), want, rtol=1e-9, atol=1e-9)
    out.backward(gy.double())
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
    if variant == "radar":
        rows, gw = 0, {}
        for i, idx in ((1, PO.SPATIAL), (2, PO.VELOCITY), (3, PO.SNR)):
            n_i = getattr(layer, f"linear{i}").out_features
            gw[f"pfn_layers.0.linear{i}.weight"] = Wt.grad[rows:rows + n_i][:, list(idx)]
            gw[f"pfn_layers.0.norm{i}.weight"], gw[f"pfn_layers.0.norm{i}.bias"] = gt.grad[rows:rows + n_i], bt.grad[rows:rows + n_i]
            rows += n_i
    else:
        gw = {"pfn_layers.0.linear.weight": Wt.grad, "pfn_layers.0.norm.weight": gt.grad, "pfn_layers.0.norm.bias": bt.grad}
    assert set(gw) == set(got)
    for n in gw:
        assert rel(got[n], gw[n]) <= 1e-4, (n, rel(got[n], gw[n]))
    n_rows = z.shape[0]
    rm_want, rv_want = 0.01 * mu.detach(), 0.99 + 0.01 * va.detach() * n_rows / (n_rows - 1)
    bufs = dict(net.named_buffers())
    names = [f"pfn_layers.0.norm{i}" for i in (1, 2, 3)] if variant == "radar" else ["pfn_layers.0.norm"]
    assert rel(torch.cat([bufs[n + ".running_mean"] for n in names]).cpu().double(), rm_want) <= 1e-5
    assert rel(torch.cat([bufs[n + ".running_var"] for n in names]).cpu().double(), rv_want) <= 1e-5
    assert all(int(bufs[n + ".num_batches_tracked"]) == 1 for n in names)
    # --- the module's own torch formulation (what runs with OMNIHD_PFN_FUSED=0 and on the CPU), fp32: same output
    monkeypatch.setenv("OMNIHD_PFN_FUSED", "0")
    ref = ref.float()
    with torch.no_grad():
        yr = ref(torch.from_numpy(vox), torch.from_numpy(num), torch.from_numpy(coors))
    assert np.abs(yr.numpy() - want).max() <= 2e-5 * scale
    # --- run-to-run identical
    monkeypatch.setenv("OMNIHD_PFN_FUSED", "1")
    net.zero_grad()
    y2 = net(t(vox, cuda), t(num, cuda), t(coors, cuda))
    y2.backward(gy.to(cuda))
    assert torch.equal(y2, y)
    assert all(torch.equal(p_.grad.cpu().double(), got[n]) for n, p_ in net.named_parameters())


def test_inference_statistics_inside_a_differentiated_graph(cuda, monkeypatch):
    """BatchNorm in eval mode while gradients are taken (fine-tuning with frozen statistics; the tiny-detector parity tests): the
    fused backward then has no batch-statistics terms.  Against the module's torch formulation in float64 on the CPU."""
    PillarFeatureNetV1, _ = _nets()
    rng = np.random.default_rng(3)
    vox, num, coors = _pillars(rng, 5000, 8)
    torch.manual_seed(2)
    net = PillarFeatureNetV1(in_channels=8, feat_channels=[64], voxel_size=VSZ, point_cloud_range=PCR, norm_cfg=NCFG)
    bn = net.pfn_layers[0].norm
    bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.2)
    W, gam, bet, rm, rv = (a.detach().clone().double() for a in (net.pfn_layers[0].linear.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var))
    net = net.to(cuda).eval()
    gy = torch.randn(len(vox), 64, generator=torch.Generator().manual_seed(1))
    y = net(t(vox, cuda), t(num, cuda), t(coors, cuda))
    y.backward(gy.to(cuda))
    Wt, gt, bt = W.requires_grad_(), gam.requires_grad_(), bet.requires_grad_()
    z = torch.from_numpy(PO.decorate(vox, num, coors, VSZ, PCR)) @ Wt.t()
    out = torch.relu((z - rm) * torch.rsqrt(rv + 1e-3) * gt + bt).max(1)[0]
    out.backward(gy.double())
    rel = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())
    assert rel(y.detach(), out.detach()) <= 1e-5
    assert rel(net.pfn_layers[0].linear.weight.grad, Wt.grad) <= 1e-4
    assert rel(bn.weight.grad, gt.grad) <= 1e-4 and rel(bn.bias.grad, bt.grad) <= 1e-4
    assert torch.equal(bn.running_mean.cpu().double(), rm) and torch.equal(bn.running_var.cpu().double(), rv)


def test_unsupported_forms_keep_the_torch_formulation(cuda):
    """Two layers, 'avg' mode or more than 16 decorated channels are not what the fused kernels cover: the module falls back
    to its torch formulation instead of failing."""
    PillarFeatureNetV1, RadarPillarFeatureNet = _nets()
    rng = np.random.default_rng(0)
    vox, num, coors = _pillars(rng, 800, 8)
    for net in (PillarFeatureNetV1(in_channels=8, feat_channels=[32, 64], voxel_size=VSZ, point_cloud_range=PCR, norm_cfg=NCFG),
                PillarFeatureNetV1(in_channels=8, feat_channels=[64], voxel_size=VSZ, point_cloud_range=PCR, norm_cfg=NCFG, mode="avg"),
                RadarPillarFeatureNet(in_channels=8, feat_channels=[64], voxel_size=VSZ, point_cloud_range=PCR, norm_cfg=NCFG)):
        net = net.to(cuda).train()
        try:
            y = net(t(vox, cuda), t(num, cuda), t(coors, cuda))
        except RuntimeError:
            continue                      # the 8-channel radar net's index sets need 16 channels: the reference fails there too
        assert y.shape == (len(vox), 64) and torch.isfinite(y).all()


def _sync_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path[:0] = [root, os.path.join(root, "omnihd-scenes_amd")]
        from projects.mmdet3d_plugin.rcfusion.voxel_encoders import PillarFeatureNetV1
        from omnihd_amd.mm.sync_bn import AllReduceSum
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(100 + rank)
        vox, num, coors = _pillars(rng, 6000 + 3000 * rank, 8)          # different pillar counts per rank
        torch.manual_seed(5)
        net = PillarFeatureNetV1(in_channels=8, feat_channels=[64], voxel_size=VSZ, point_cloud_range=PCR, norm_cfg=NCFG)
        net.pfn_layers[0].norm.weight.data = torch.rand(64) + 0.5
        net.pfn_layers[0].norm.bias.data = torch.randn(64) * 0.2
        layer = net.pfn_layers[0]
        lin_w, gam, bet = (p_.detach().clone().double() for p_ in (layer.linear.weight, layer.norm.weight, layer.norm.bias))
        net = net.to(dev).train()
        gy = torch.randn(len(vox), 64, generator=torch.Generator().manual_seed(9 + rank))
        y = net(t(vox, dev), t(num, dev), t(coors, dev))
        y.backward(gy.to(dev))
        got = dict(y=y.detach().cpu().double(), gw=net.pfn_layers[0].linear.weight.grad.cpu().double(),
                   gg=net.pfn_layers[0].norm.weight.grad.cpu().double(), gb=net.pfn_layers[0].norm.bias.grad.cpu().double(),
                   rm=net.pfn_layers[0].norm.running_mean.cpu().double(), rv=net.pfn_layers[0].norm.running_var.cpu().double())
        # the reference algorithm (ops/norm.py:55-82 for the 3-D case; the same for naiveSyncBN1d upstream) in float64
        x = torch.from_numpy(PO.decorate(vox, num, coors, VSZ, PCR))
        lw, g_, b_ = lin_w.requires_grad_(), gam.requires_grad_(), bet.requires_grad_()
        z = (x @ lw.t()).reshape(-1, 64)
        vec = AllReduceSum.apply(torch.cat([z.mean(0), (z * z).mean(0)])) * (1.0 / world)
        mean, msq = vec[:64], vec[64:]
        var = msq - mean * mean
        out = torch.relu((z - mean) * torch.rsqrt(var + 1e-3) * g_ + b_).reshape(len(vox), -1, 64).max(1)[0]
        out.backward(gy.double())
        want = dict(y=out.detach(), gw=lw.grad, gg=g_.grad, gb=b_.grad, rm=0.01 * mean.detach(), rv=0.99 + 0.01 * var.detach())
        q.put((rank, {k: float((got[k] - want[k]).norm() / (want[k].norm() + 1e-30)) for k in got}))
    finally:
        dist.destroy_process_group()


def test_rank_averaged_statistics_two_ranks_on_one_gpu(cuda):
    """naiveSyncBN1d over two ranks (gloo, both on this GPU, different pillar counts): the fused path exchanges the K + K*K
    moments forward and [sum g | sum g*yhat] backward; output, all three gradients and the running statistics (naive update with
    the biased variance) against the reference algorithm in float64."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + os.getpid() % 600
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, errs = q.get(timeout=120)
        res[rank] = errs
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, errs in res.items():
        assert errs["y"] < 1e-5 and errs["rm"] < 1e-5 and errs["rv"] < 1e-5, (rank, errs)
        assert errs["gw"] < 1e-4 and errs["gg"] < 1e-4 and errs["gb"] < 1e-4, (rank, errs)


def test_scatter_with_the_persistent_cell_map(cuda):
    """ops.pillar_scatter keeps one cell map per grid that is clean between calls (the canvas kernel resets what the map kernel
    entered): repeated calls with different pillar sets, both layouts, duplicates (last pillar wins) — bit-exact vs the oracle."""
    from omnihd_amd import ops
    from oracle import cpu as OC
    rng = np.random.default_rng(4)
    for it in range(4):
        m = int(rng.integers(500, 2000))
        cells = rng.choice(40 * 56, m - 5, replace=False)
        cells = np.concatenate([cells, cells[:5]])                               # five duplicated cells
        coors = np.stack([rng.integers(0, 2, m), np.zeros(m, np.int64), cells // 56, cells % 56], 1).astype(np.int32)
        feats = rng.standard_normal((m, 64), dtype=np.float32)
        want = OC.pillar_scatter(feats, coors, 2, 40, 56)
        for cl in (True, False):
            got = ops.pillar_scatter(t(feats, cuda), t(coors, cuda), 2, 40, 56, channels_last=cl)
            assert got.shape == (2, 64, 40, 56)
            assert np.array_equal(got.cpu().numpy(), want), (it, cl)
    for ent in ops._SCATTER_MAPS.values():
        assert not ent[1] and int((ent[0] != -1).sum()) == 0                      # every map is clean again
