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
