"""CPU tests of the point-stream-only detectors: ``MVXFasterRCNN`` as a registered detector (the
reference's radar-only / LiDAR-only PointPillars configs) and the upstream ``HardVFE`` voxel encoder.
HIP operators are routed to the CPU oracle (tests only); tests/test_pillars_gpu.py repeats the step
through the HIP path."""
import math
import os

import numpy as np
import pytest
import torch

REF_CFGS = "/root/reference/projects/configs/"
STREAM_CONFIGS = [("bevfusion_NewScenes/radar_stream/pointpillars_4DRadar.py", "radar", 4853560),
                  ("PointPillars_NewScenes/pointpillars_4DRadar.py", "radar", 4853560),
                  ("RCFusion_NewScenes/radar_stream/RadarPillarNet.py", "rcfusion", 4853112),
                  ("PointPillars_NewScenes/pointpillars_LiDAR.py", "lidar", 4861688)]


@pytest.mark.parametrize("path,stream,n_params", STREAM_CONFIGS)
def test_reference_stream_configs_build_unchanged(path, stream, n_params):
    """type='MVXFasterRCNN' resolves; the restated dict used off the authoring machine is the reference's."""
    from omnihd_amd import harness
    from omnihd_amd.mm.config import build_detector, load_config
    cfg = harness.pillars_model_cfg(harness.reference_model_cfg(), stream)
    if os.path.exists(REF_CFGS + path):
        assert load_config(REF_CFGS + path)["model"] == cfg
    m = build_detector(cfg)
    assert type(m).__name__ == "MVXFasterRCNN" and sum(p.numel() for p in m.parameters()) == n_params
    keys = set(m.state_dict())
    assert {"pts_backbone.blocks.0.0.weight", "pts_neck.deblocks.1.0.weight", "pts_bbox_head.conv_reg.weight"} <= keys
    assert not any(k.startswith(("img_", "lift_splat_shot_vis", "reduc_conv")) for k in keys)
    if stream == "lidar":
        assert {"pts_voxel_encoder.vfe_layers.0.linear.weight", "pts_voxel_encoder.vfe_layers.1.norm.running_var"} <= keys
        assert m.pts_voxel_encoder.vfe_layers[0].linear.weight.shape == (64, 10)
        assert m.pts_voxel_encoder.vfe_layers[1].linear.weight.shape == (64, 128)
        assert m.pts_voxel_layer.max_num_points == 64


def test_every_non_bevformer_reference_config_builds():
    import glob
    from omnihd_amd.mm.config import build_detector, load_config
    if not os.path.isdir(REF_CFGS):
        pytest.skip("reference checkout not on this machine")
    files = [f for f in glob.glob(REF_CFGS + "*/*.py") + glob.glob(REF_CFGS + "*/*/*.py")
             if "_base_" not in f and "/datasets/" not in f and "bevformer" not in f]
    assert len(files) == 9
    for f in files:
        assert build_detector(load_config(f)["model"]) is not None, f


def test_vfe_layer_known_answers():
    """Identity-like weights and an identity BatchNorm (eval, mean 0, var 1-eps): values by hand."""
    from omnihd_amd.mm.hard_vfe import VFELayer
    x = torch.tensor([[[1.0, -2.0], [3.0, 0.5], [0.0, 0.0]],
                      [[-1.0, -1.0], [-4.0, 2.0], [0.0, 0.0]]])
    for cat_max, max_out in [(True, True), (False, True), (False, False)]:
        lay = VFELayer(2, 2, norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), max_out=max_out, cat_max=cat_max).eval()
        with torch.no_grad():
            lay.linear.weight.copy_(torch.tensor([[1.0, 0.0], [0.0, 2.0]]))
            lay.norm.running_var.fill_(1.0 - 1e-3)       # so that x / sqrt(var + eps) == x
        y = lay(x)
        point = torch.tensor([[[1.0, 0.0], [3.0, 1.0], [0.0, 0.0]], [[0.0, 0.0], [0.0, 4.0], [0.0, 0.0]]])
        agg = torch.tensor([[3.0, 1.0], [0.0, 4.0]])
        if not max_out:
            want = point
        elif not cat_max:
            want = agg
        else:
            want = torch.cat([point, agg[:, None, :].expand(2, 3, 2)], dim=2)
        assert y.shape == want.shape and torch.allclose(y, want, atol=1e-6), (cat_max, max_out)


def test_hard_vfe_decorations_and_padding_known_answers():
    """One layer with a 10x10 identity: the output is the max over slots of relu(decorated features)."""
    from omnihd_amd.mm.hard_vfe import HardVFE
    vs, pcr = [0.5, 0.5, 2.0], [-8.0, -6.0, -1.0, 8.0, 6.0, 1.0]
    net = HardVFE(in_channels=4, feat_channels=[10], with_cluster_center=True, with_voxel_center=True, voxel_size=vs,
                  point_cloud_range=pcr, norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01)).eval()
    assert net.in_channels == 10 and net.num_vfe == 1
    with torch.no_grad():
        net.vfe_layers[0].linear.weight.copy_(torch.eye(10))
        net.vfe_layers[0].norm.running_var.fill_(1.0 - 1e-3)
    vox = torch.zeros(2, 3, 4)
    vox[0, 0] = torch.tensor([1.1, 2.2, 0.3, 0.9])
    vox[0, 1] = torch.tensor([1.3, 2.4, -0.1, 0.5])
    vox[0, 2] = torch.tensor([9.0, 9.0, 9.0, 9.0])         # beyond num_points: must be masked out
    vox[1, 0] = torch.tensor([-7.9, -5.8, 0.0, 0.2])
    nump = torch.tensor([2, 1], dtype=torch.int32)
    coors = torch.tensor([[0, 0, 16, 18], [1, 0, 0, 0]], dtype=torch.int32)     # [batch, z, y, x]
    y = net(vox, nump, coors)
    # voxel 0: mean over the THREE slots' sum / 2 points (upstream sums the padded slot too, it is zero in real
    # voxeliser output; here slot 2 is poisoned, so the cluster mean includes it — then the slot is masked)
    mean0 = (vox[0, :, :3].sum(0)) / 2.0
    centre0 = torch.tensor([18 * 0.5 + (-8.0 + 0.25), 16 * 0.5 + (-6.0 + 0.25), 0 * 2.0 + (-1.0 + 1.0)])
    rows0 = torch.cat([vox[0, :2], vox[0, :2, :3] - mean0, vox[0, :2, :3] - centre0], dim=1)
    want0 = torch.relu(rows0).max(0)[0]
    centre1 = torch.tensor([-7.75, -5.75, 0.0])
    row1 = torch.cat([vox[1, 0], vox[1, 0, :3] - vox[1, 0, :3], vox[1, 0, :3] - centre1])
    want1 = torch.relu(row1)
    assert torch.allclose(y[0], torch.maximum(want0, torch.zeros(10)), atol=1e-5)
    assert torch.allclose(y[1], want1, atol=1e-5)
    for bad in (dict(with_distance=True), dict(fusion_layer=dict(type="PointFusion")), dict(return_point_feats=True)):
        with pytest.raises(NotImplementedError):
            HardVFE(in_channels=4, feat_channels=[8], **bad)


def test_hard_vfe_padded_slots_take_part_in_the_max_as_upstream():
    """After Linear(no bias) -> BN -> ReLU a zeroed slot carries relu(beta - mean*scale): upstream keeps it."""
    from omnihd_amd.mm.hard_vfe import HardVFE
    net = HardVFE(in_channels=4, feat_channels=[4], voxel_size=[1, 1, 1], point_cloud_range=[0, 0, 0, 4, 4, 1]).eval()
    with torch.no_grad():
        net.vfe_layers[0].linear.weight.copy_(torch.eye(4))
        net.vfe_layers[0].norm.running_var.fill_(1.0 - 1e-3)
        net.vfe_layers[0].norm.bias.copy_(torch.tensor([0.0, 5.0, 0.0, 0.0]))
    vox = torch.zeros(1, 2, 4)
    vox[0, 0] = torch.tensor([1.0, -9.0, 2.0, 3.0])
    y = net(vox, torch.tensor([1], dtype=torch.int32), torch.zeros(1, 4, dtype=torch.int32))
    assert torch.allclose(y[0], torch.tensor([1.0, 5.0, 2.0, 3.0]), atol=1e-5)   # 5.0 comes from the EMPTY slot


@pytest.mark.parametrize("stream", ["radar", "rcfusion", "lidar"])
def test_stream_only_detector_trains_and_tests_a_tiny_step_over_the_oracle(stream):
    from omnihd_amd import harness
    from omnihd_amd.mm.config import build_detector
    from oracle.torch_shim import oracle_ops
    torch.set_num_threads(4)
    torch.manual_seed(0)
    with oracle_ops():
        model = build_detector(harness.pillars_model_cfg(harness.tiny_model_cfg(7), stream))
        b = harness.synthetic_batch("tiny", 2, 7, "cpu", 0)
        pts = [p[:, :4].contiguous() for p in b["points"]] if stream == "lidar" else b["points"]
        model.train()
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=2e-3)
        hist = []
        for _ in range(4):
            losses = model(return_loss=True, points=pts, img_metas=b["img_metas"], gt_bboxes_3d=b["gt_bboxes_3d"],
                           gt_labels_3d=b["gt_labels_3d"])
            total = sum(v[0] if isinstance(v, list) else v for v in losses.values())
            opt.zero_grad()
            total.backward()
            opt.step()
            hist.append(float(total.detach()))
        assert set(losses) == {"loss_cls", "loss_bbox", "loss_dir"}
        assert all(math.isfinite(h) for h in hist) and hist[-1] < hist[0]
        assert all(p.grad is not None for p in model.pts_voxel_encoder.parameters())
        model.eval()
        torch.nn.init.constant_(model.pts_bbox_head.conv_cls.bias, 0.0)
        out = model(return_loss=False, points=[pts], img_metas=[b["img_metas"]])
    assert len(out) == 2 and all("pts_bbox" in r for r in out)
    assert all(len(r["pts_bbox"]["boxes_3d"]) == r["pts_bbox"]["scores_3d"].numel() for r in out)
    assert np.isfinite(hist).all()


@pytest.mark.parametrize("norm", ["BN1d", "naiveSyncBN1d"])
@pytest.mark.parametrize("chans", [[64, 64], [32], [16, 24, 40]])
def test_hard_vfe_packed_evaluation_equals_the_dense_one(norm, chans):
    """Real points + one representative row per voxel (standing for its empty slots in the max and, weighted, in the
    BatchNorm statistics) == all M x T slots: outputs, parameter gradients and running statistics, eval and train."""
    import copy
    from omnihd_amd.mm.hard_vfe import HardVFE
    torch.manual_seed(len(chans) * 7 + len(norm))
    M, T = 150, 16
    dense = HardVFE(in_channels=4, feat_channels=chans, with_cluster_center=True, with_voxel_center=True,
                    voxel_size=[0.5, 0.5, 2.0], point_cloud_range=[-8.0, -6.0, -1.0, 8.0, 6.0, 1.0],
                    norm_cfg=dict(type=norm, eps=1e-3, momentum=0.01), packed=False)
    for m in dense.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.normal_(); m.running_var.uniform_(0.5, 2); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.5)
    packed = copy.deepcopy(dense)
    packed.packed = True
    n = torch.randint(1, T + 1, (M,), dtype=torch.int32)
    n[:5] = T                                                   # full voxels: no empty slot takes part
    n[5:10] = 1
    vox = torch.zeros(M, T, 4)
    for k in range(M):
        vox[k, :n[k]] = torch.randn(int(n[k]), 4) * 3
    coors = torch.stack([torch.zeros(M), torch.zeros(M), torch.randint(0, 24, (M,)), torch.randint(0, 32, (M,))], 1).int()
    for mode in ("eval", "train"):
        getattr(dense, mode)(); getattr(packed, mode)()
        a = dense(vox, n, coors)
        b = packed(vox, n, coors, max_real_points=int(n.sum()) + 11)          # any upper bound of the real slots
        assert a.shape == b.shape == (M, chans[-1]) and float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), mode
        w = torch.randn_like(a)
        dense.zero_grad(); packed.zero_grad()
        (a * w).sum().backward(); (b * w).sum().backward()
        for (name, p), q in zip(dense.named_parameters(), packed.parameters()):
            assert float((p.grad - q.grad).abs().max()) <= 2e-5 * float(p.grad.abs().max() + 1e-9), (mode, name)
        for (name, x), y in zip(dense.named_buffers(), packed.buffers()):
            assert torch.allclose(x.float(), y.float(), atol=2e-6), (mode, name)
    # without the hint the buffer is sized by the slots (still exact), and the switch can come from the environment
    assert torch.allclose(packed(vox, n, coors), dense(vox, n, coors), atol=1e-4)


def test_packed_encoder_on_voxeliser_output_matches_a_float64_yardstick(monkeypatch):
    """On real voxeliser output (64 slots, 1-5 points per pillar: 95 % empty slots) the two evaluations are the same
    function (1e-13 in float64); in float32 the packed one is the ACCURATE one — the dense BatchNorm backward cancels
    over 38 000 mostly-zero rows and is off by ~1 % — so both are held against the float64 dense result."""
    import copy
    from omnihd_amd import harness
    from omnihd_amd.mm.config import build_detector
    from oracle.torch_shim import oracle_ops
    torch.set_num_threads(4)
    with oracle_ops():
        torch.manual_seed(0)
        m = build_detector(harness.pillars_model_cfg(harness.tiny_model_cfg(7), "lidar")).train()
        b = harness.synthetic_batch("tiny", 2, 7, "cpu", 0)
        pts = [p[:, :4].contiguous() for p in b["points"]]
        vox, num, coors = m.voxelize(pts)
        hints = m._encoder_hints(pts)
        assert hints == dict(max_real_points=sum(p.shape[0] for p in pts)) and int(num.sum()) <= hints["max_real_points"]
        # whole-detector step with the switch on: same loss as with it off
        totals = {}
        for flag in ("0", "1"):
            monkeypatch.setenv("OMNIHD_VFE_PACKED", flag)
            torch.manual_seed(0)
            d = build_detector(harness.pillars_model_cfg(harness.tiny_model_cfg(7), "lidar")).train()
            losses = d(return_loss=True, points=pts, img_metas=b["img_metas"], gt_bboxes_3d=b["gt_bboxes_3d"],
                       gt_labels_3d=b["gt_labels_3d"])
            totals[flag] = float(sum(v[0] if isinstance(v, list) else v for v in losses.values()).detach())
        assert abs(totals["0"] - totals["1"]) <= 1e-5 * abs(totals["0"])
    w = torch.randn(vox.shape[0], 64, generator=torch.Generator().manual_seed(1))

    def run(dtype, packed):
        e = copy.deepcopy(m.pts_voxel_encoder).to(dtype)
        e.packed = packed
        out = e(vox.to(dtype), num, coors, **hints)
        (out * w.to(dtype)).sum().backward()
        return out.detach().double(), e.vfe_layers[0].linear.weight.grad.double(), e.vfe_layers[1].norm.running_var.double()

    rel = lambda a, ref: float((a - ref).abs().max() / ref.abs().max())       # noqa: E731
    d64, p64, d32, p32 = run(torch.float64, False), run(torch.float64, True), run(torch.float32, False), run(torch.float32, True)
    assert all(rel(a, r) < 1e-11 for a, r in zip(p64, d64))                     # the same function
    assert rel(p32[0], d64[0]) < 1e-5 and rel(p32[1], d64[1]) < 1e-4 and rel(p32[2], d64[2]) < 1e-5
    assert rel(d32[0], d64[0]) < 1e-4 and rel(d32[1], d64[1]) < 5e-2           # the dense float32 gradient is the loose one
