"""CPU tests of the host-side detector code: registry/config boundary, layer mirrors against golden
vectors captured from the reference, known-answer cases for the upstream pieces the reference does
not vendor (anchors, box coder, assigner), and one training step of a scaled-down detector with the
HIP ops routed to the CPU oracle (tests only)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def test_reference_config_builds_unchanged_with_reference_state_dict_keys():
    from omnihd_amd import harness
    from omnihd_amd.mm.config import build_detector, load_config
    ref = "/root/reference/projects/configs/bevfusion_NewScenes/bevfusion.py"
    cfg = load_config(ref)["model"] if os.path.exists(ref) else harness.reference_model_cfg()
    if os.path.exists(ref):   # the restated dict used on machines without the reference must be identical
        assert cfg == harness.reference_model_cfg()
    m = build_detector(cfg)
    keys = set(m.state_dict())
    for k in ["lift_splat_shot_vis.frustum", "lift_splat_shot_vis.bevencode.0.weight", "lift_splat_shot_vis.bevencode.9.weight",
              "lift_splat_shot_vis.bevencode.10.running_mean", "lift_splat_shot_vis.camencode.depthnet.reduce_conv.0.weight",
              "lift_splat_shot_vis.camencode.depthnet.depth_conv.3.aspp2.atrous_conv.weight",
              "lift_splat_shot_vis.camencode.depthnet.depth_conv.4.conv_offset.weight",
              "lift_splat_shot_vis.camencode.depthnet.depth_conv.5.bias", "reduc_conv.conv.weight", "reduc_conv.bn.weight",
              "seblock.att.1.weight", "pts_voxel_encoder.pfn_layers.0.linear.weight", "pts_voxel_encoder.pfn_layers.0.norm.weight",
              "pts_backbone.blocks.2.15.weight", "pts_neck.deblocks.2.0.weight", "img_backbone.layer4.2.conv3.weight",
              "img_neck.lateral_convs.0.conv.weight", "img_neck.fpn_convs.2.conv.weight", "img_neck.adp.1.1.conv.weight",
              "img_neck.reduc_conv.conv.weight", "pts_bbox_head.conv_cls.weight", "pts_bbox_head.conv_dir_cls.bias"]:
        assert k in keys, k
    assert m.pts_bbox_head.conv_cls.out_channels == 32 and m.pts_bbox_head.conv_reg.out_channels == 72
    assert m.lift_splat_shot_vis.D == 59 and m.lift_splat_shot_vis.nx.tolist() == [240, 160, 16]
    assert sum(p.numel() for p in m.parameters()) == 66094341
    frozen = [n for n, p in m.named_parameters() if not p.requires_grad]
    assert any(n.startswith("img_backbone.layer1") for n in frozen)
    assert all("img_backbone" in n or n == "lift_splat_shot_vis.frustum" for n in frozen)


def test_registry_type_names_of_the_plugin():
    import projects.mmdet3d_plugin  # noqa: F401
    from omnihd_amd.mm import DETECTORS, NECKS, NORM_LAYERS, VOXEL_ENCODERS
    assert "BEVFUSION_depth" in DETECTORS and "FPNC" in NECKS
    assert "PillarFeatureNetV1" in VOXEL_ENCODERS and "RadarPillarFeatureNet" in VOXEL_ENCODERS
    for n in ("naiveSyncBN1d", "naiveSyncBN2d", "naiveSyncBN3d"):
        assert n in NORM_LAYERS


def test_gaussian_depth_target_matches_reference(golden):
    from projects.mmdet3d_plugin.utils.gaussian import generate_guassian_depth_target
    t, m = generate_guassian_depth_target(torch.from_numpy(golden["g5_depth_map"]), 4, [1.0, 9.0, 1.0], constant_std=0.5)
    np.testing.assert_allclose(t.numpy(), golden["g5_target"], rtol=0, atol=1e-6)
    assert np.array_equal(m.numpy(), golden["g5_min_depth"])


def test_pillar_feature_nets_match_reference(golden):
    from projects.mmdet3d_plugin.rcfusion.voxel_encoders import PillarFeatureNetV1, RadarPillarFeatureNet
    vsz, pcr = [0.25, 0.25, 8], [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]
    ncfg = dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01)
    net = PillarFeatureNetV1(in_channels=8, feat_channels=[64], with_distance=False, voxel_size=vsz, point_cloud_range=pcr, norm_cfg=ncfg)
    net.pfn_layers[0].linear.weight.data = torch.from_numpy(golden["g6_pfn_linear_w"])
    bn = net.pfn_layers[0].norm
    w, b, rm, rv = [torch.from_numpy(x) for x in golden["g6_pfn_bn"]]
    bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var = w, b, rm, rv
    net.eval()
    vox = torch.from_numpy(golden["g6_voxels"])
    keep = vox.clone()
    with torch.no_grad():
        y = net(vox, torch.from_numpy(golden["g6_num_points"]), torch.from_numpy(golden["g6_coors"]))
    np.testing.assert_allclose(y.numpy(), golden["g6_pfn_out"], rtol=1e-5, atol=1e-5)
    assert torch.equal(vox, keep)           # unlike the reference (legacy=True), the caller's tensor is not mutated
    rnet = RadarPillarFeatureNet(in_channels=7, feat_channels=[64], with_distance=False, voxel_size=vsz, point_cloud_range=pcr, norm_cfg=ncfg)
    sd = {k[len("g6_radar_sd__"):].replace("__", "."): torch.from_numpy(golden[k]) for k in golden.files if k.startswith("g6_radar_sd__")}
    assert not rnet.load_state_dict(sd, strict=False).unexpected_keys      # reference state-dict keys load
    rnet.eval()
    with torch.no_grad():
        y7 = rnet(vox[:, :, :7].clone(), torch.from_numpy(golden["g6_num_points"]), torch.from_numpy(golden["g6_coors"]))
    np.testing.assert_allclose(y7.numpy(), golden["g6_radar_out"], rtol=1e-5, atol=1e-5)


def test_lss_geometry_and_frustum_match_reference(golden):
    from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet import LiftSplatShoot_Depth, gen_dx_bx
    dx, bx, nx = gen_dx_bx([-60.0, 60.0, 0.5], [-40.0, 40.0, 0.5], [-3.0, 5.0, 0.5])
    assert np.array_equal(dx.numpy(), golden["g1_r1_dx"]) and np.array_equal(bx.numpy(), golden["g1_r1_bx"])
    assert np.array_equal(nx.numpy(), golden["g1_r1_nx"])
    lss = LiftSplatShoot_Depth(final_dim=(32, 48), camera_depth_range=[1.0, 9.0, 1.0], pc_range=golden["g2_pc_range"].tolist(),
                               downsample=4, grid=1.0, inputC=16, camC=8, norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01))
    fr = lss.frustum.numpy()
    assert np.array_equal(fr[0, 0, :, 0], golden["g1_tiny_xs"]) and np.array_equal(fr[0, :, 0, 1], golden["g1_tiny_ys"])
    geom = lss.get_geometry(torch.from_numpy(golden["g2_rots"]), torch.from_numpy(golden["g2_trans"]))
    assert np.array_equal(geom.numpy(), golden["g2_geom"])      # bit for bit the reference's torch-CPU geometry
    assert [k for k, _ in lss.bevencode.named_parameters()][:2] == ["0.weight", "1.weight"]


def test_lss_geometry_is_fp32_under_autocast_and_general_path_equals_the_reference_formula(golden):
    """bf16 autocast must not touch the geometry (a matmul formulation is down-cast by torch and moves points across
    voxel borders); post-/extra- transforms give what the reference's batched-matmul text gives."""
    from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet import LiftSplatShoot_Depth
    lss = LiftSplatShoot_Depth(final_dim=(32, 48), camera_depth_range=[1.0, 9.0, 1.0], pc_range=golden["g2_pc_range"].tolist(),
                               downsample=4, grid=1.0, inputC=16, camC=8, norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01))
    rots, trans = torch.from_numpy(golden["g2_rots"]), torch.from_numpy(golden["g2_trans"])
    with torch.autocast("cpu", dtype=torch.bfloat16):
        geom = lss.get_geometry(rots, trans)
    assert geom.dtype == torch.float32 and np.array_equal(geom.numpy(), golden["g2_geom"])
    B, N = rots.shape[:2]
    torch.manual_seed(0)
    post_rots = torch.linalg.qr(torch.randn(B, N, 3, 3))[0] + 0.1 * torch.randn(B, N, 3, 3)
    extra_rots = torch.linalg.qr(torch.randn(B, N, 3, 3))[0]
    post_trans, extra_trans = torch.randn(B, N, 3), torch.randn(B, N, 3)
    # the reference's text (cam_stream_lss_bevpoolv2_depthnet.py:244-263) with all four optional transforms
    p = lss.frustum.data - post_trans.view(B, N, 1, 1, 1, 3)
    p = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3).matmul(p.unsqueeze(-1))
    p = torch.cat((p[:, :, :, :, :, :2] * p[:, :, :, :, :, 2:3], p[:, :, :, :, :, 2:3]), 5)
    p = rots.view(B, N, 1, 1, 1, 3, 3).matmul(p).squeeze(-1) + trans.view(B, N, 1, 1, 1, 3)
    p = extra_rots.view(B, N, 1, 1, 1, 3, 3).matmul(p.unsqueeze(-1)).squeeze(-1) + extra_trans.view(B, N, 1, 1, 1, 3)
    got = lss.get_geometry(rots, trans, post_rots, post_trans, extra_rots, extra_trans)
    assert got.shape == p.shape and float((got - p).abs().max()) <= 1e-5 * float(p.abs().max())
    only_extra = lss.get_geometry(rots, trans, extra_trans=extra_trans)
    assert torch.equal(only_extra, lss.get_geometry(rots, trans) + extra_trans.view(B, N, 1, 1, 1, 3))


def test_dcn_zero_offsets_is_a_grouped_conv_and_offsets_shift_samples():
    from omnihd_amd.mm.dcn import DeformConv2dPack
    torch.manual_seed(0)
    m = DeformConv2dPack(16, 16, 3, padding=1, groups=4)
    x = torch.randn(2, 16, 9, 11)
    torch.testing.assert_close(m(x), F.conv2d(x, m.weight, None, 1, 1, 1, 4), rtol=1e-5, atol=1e-5)
    # a constant offset of (+1 row, 0) for every tap == convolving the image shifted up by one row
    m.conv_offset.bias.data.view(9, 2)[:, 0] = 1.0
    shifted = torch.cat([x[:, :, 1:], torch.zeros_like(x[:, :, :1])], dim=2)
    # (output row 0 differs by construction: its top taps sample the real row 0, not the conv's zero padding)
    torch.testing.assert_close(m(x)[:, :, 1:], F.conv2d(shifted, m.weight, None, 1, 1, 1, 4)[:, :, 1:], rtol=1e-4, atol=1e-4)


def test_anchor_generator_order_and_centres():
    from omnihd_amd.harness import ANCHOR_SIZES, ANCHOR_Z
    from omnihd_amd.mm.anchor_head import AlignedAnchor3DRangeGenerator
    gen = AlignedAnchor3DRangeGenerator(ranges=[[-60, -40, z, 60, 40, z] for z in ANCHOR_Z], sizes=ANCHOR_SIZES,
                                        custom_values=[0, 0], rotations=[0, 1.57], reshape_out=True)
    a = gen.grid_anchors([(160, 240)], device="cpu")[0]
    assert a.shape == (160 * 240 * 8, 9) and gen.num_base_anchors == 8
    a = a.view(160, 240, 4, 2, 9)
    np.testing.assert_allclose(a[0, 0, 0, 0, :2].numpy(), [-59.75, -39.75], atol=1e-5)     # cell centres
    np.testing.assert_allclose(a[159, 239, 0, 0, :2].numpy(), [59.75, 39.75], atol=1e-4)
    for s in range(4):
        np.testing.assert_allclose(a[3, 7, s, 1, 3:6].numpy(), ANCHOR_SIZES[s], rtol=1e-6)
        assert abs(float(a[3, 7, s, 0, 2]) - ANCHOR_Z[s]) < 1e-6
    assert float(a[0, 0, 0, 0, 6]) == 0.0 and abs(float(a[0, 0, 0, 1, 6]) - 1.57) < 1e-6
    assert a[..., 7:].abs().sum() == 0


def test_box_coder_known_answer_and_roundtrip():
    from omnihd_amd.mm.anchor_head import DeltaXYZWLHRBBoxCoder
    anchor = torch.tensor([[0.0, 0.0, -1.0, 2.0, 4.0, 1.5, 0.0, 0.0, 0.0]])
    gt = torch.tensor([[1.0, -2.0, -0.5, 2.2, 4.4, 1.8, 0.3, 1.0, -1.0]])
    code = DeltaXYZWLHRBBoxCoder(code_size=9).encode(anchor, gt)
    diag = math.sqrt(4.0 ** 2 + 2.0 ** 2)
    want = [1 / diag, -2 / diag, ((-0.5 + 0.9) - (-1.0 + 0.75)) / 1.5, math.log(1.1), math.log(1.1), math.log(1.2), 0.3, 1.0, -1.0]
    np.testing.assert_allclose(code[0].numpy(), want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(DeltaXYZWLHRBBoxCoder.decode(anchor, code), gt, rtol=1e-5, atol=1e-5)


def test_max_iou_assigner_known_answer():
    from omnihd_amd.mm.anchor_head import MaxIoUAssigner
    asg = MaxIoUAssigner(pos_iou_thr=0.6, neg_iou_thr=0.3, min_pos_iou=0.3, iou_calculator=dict(type="BboxOverlapsNearest3D"))

    def box(x, y, w, l, yaw=0.0):
        return [x, y, 0, w, l, 1, yaw, 0, 0]
    anchors = torch.tensor([box(0, 0, 2, 4), box(0.2, 0, 2, 4), box(10, 10, 2, 4), box(1.0, 0, 2, 4), box(5, 5, 4, 2, 1.57), box(0, 3.0, 2, 4)])
    gts = torch.tensor([box(0, 0, 2, 4), box(5, 5, 2, 4, 0.0), box(0, 2.2, 2, 4)])
    got = asg.assign(anchors, gts).tolist()
    # a0: IoU 1 with gt0 -> 1; a1: IoU 0.818 -> 1; a2: no overlap -> 0 (negative); a3: IoU 1/3 -> ignore (-1);
    # a4: rotated 90deg anchor vs axis-aligned gt1: IoU 4/12 = 0.33 but it is gt1's best (>= min_pos_iou) -> 2;
    # a5: IoU with gt2 = (2*3.2)/(8+8-6.4) = 0.667 -> 3
    assert got == [1, 1, 0, -1, 2, 3]
    assert asg.assign(anchors, torch.zeros(0, 9)).tolist() == [0] * 6


def test_focal_and_smooth_l1_losses_match_formulas():
    from omnihd_amd.mm.anchor_head import FocalLoss, SmoothL1Loss
    torch.manual_seed(1)
    pred = torch.randn(7, 4)
    label = torch.tensor([0, 1, 4, 4, 2, 3, 4])
    w = torch.tensor([1., 1., 1., 0., 1., 1., 1.])
    t = F.one_hot(label, 5)[:, :4].float()
    p = pred.sigmoid()
    ce = -(t * torch.log(p) + (1 - t) * torch.log(1 - p))
    want = (ce * (0.25 * t + 0.75 * (1 - t)) * ((1 - p) * t + p * (1 - t)) ** 2 * w[:, None]).sum() / 3.0
    torch.testing.assert_close(FocalLoss(gamma=2.0, alpha=0.25)(pred, label, w, avg_factor=3.0), want, rtol=1e-5, atol=1e-6)
    d = torch.tensor([[0.05, 0.5]])
    torch.testing.assert_close(SmoothL1Loss(beta=1 / 9)(d, torch.zeros_like(d), torch.ones_like(d), avg_factor=1.0),
                               torch.tensor(0.5 * 0.05 ** 2 * 9 + 0.5 - 0.5 / 9))


def test_tiny_detector_training_step_on_cpu_with_oracle_ops():
    from omnihd_amd.harness import FusionTrainStep
    from oracle.torch_shim import oracle_ops
    torch.set_num_threads(4)
    with oracle_ops():
        st = FusionTrainStep(res="tiny", batch=2, radar_dims=7, device="cpu", dtype="fp32", channels_last=False, sets=1)
        l0 = float(st.step().detach())
        missing = [n for n, p in st.raw_model.named_parameters() if p.requires_grad and p.grad is None]
        for _ in range(3):
            l1 = float(st.step().detach())
    assert set(st.last_losses) == {"loss_cls", "loss_bbox", "loss_dir", "img_depth_loss"}
    assert not missing, missing
    assert math.isfinite(l0) and math.isfinite(l1) and l1 < l0


def test_detector_refuses_cpu_tensors_without_the_oracle_shim():
    from omnihd_amd.harness import FusionTrainStep
    st = FusionTrainStep(res="tiny", batch=1, radar_dims=7, device="cpu", dtype="fp32", channels_last=False, sets=1)
    with pytest.raises(RuntimeError, match="no CPU p"):
        st.step()


def test_rcfusion_config_builds_and_trains_tiny_step():
    """the RCFusion variant (RadarPillarFeatureNet + Cross_Modal_Fusion)."""
    from omnihd_amd import harness
    from omnihd_amd.mm.config import build_detector, load_config
    from oracle.torch_shim import oracle_ops
    ref = "/root/reference/projects/configs/RCFusion_NewScenes/rcfusion_lss.py"
    if os.path.exists(ref):
        m = build_detector(load_config(ref)["model"])
        keys = set(m.state_dict())
        for k in ["cross_attention.att_img.0.weight", "cross_attention.att_radar.0.weight", "cross_attention.reduce_mixBEV.conv.weight",
                  "cross_attention.reduce_mixBEV.bn.weight", "pts_voxel_encoder.pfn_layers.0.linear1.weight",
                  "pts_voxel_encoder.pfn_layers.0.norm3.weight"]:
            assert k in keys, k
        assert "reduc_conv.conv.weight" not in keys
    c = harness.tiny_model_cfg(7)
    c["type"] = "RCFusion_FasterRCNN"
    c.pop("lc_fusion")
    c["rc_fusion"] = "cross_attention"
    c["pts_voxel_encoder"].update(type="RadarPillarFeatureNet", with_velocity_snr_center=True)
    torch.set_num_threads(4)
    with oracle_ops():
        model = build_detector(c)
        model.train()
        losses = model(return_loss=True, **harness.synthetic_batch("tiny", 2, 7, "cpu", 0))
        total = sum(v[0] if isinstance(v, list) else v for v in losses.values())
        total.backward()
    assert set(losses) == {"loss_cls", "loss_bbox", "loss_dir", "img_depth_loss"} and math.isfinite(float(total.detach()))
    assert model.cross_attention.att_img[0].weight.grad is not None


def test_bilinear_resize_equals_interpolate():
    from omnihd_amd.mm.bricks import BilinearResize
    x = torch.randn(2, 3, 8, 22)
    for size in [(64, 176), (16, 44), (8, 22), (5, 7)]:
        torch.testing.assert_close(BilinearResize(size)(x), F.interpolate(x, size=size, mode="bilinear", align_corners=True),
                                   rtol=1e-5, atol=1e-5)


def test_dcn_gather_formulation_matches_grid_sample_formulation():
    from omnihd_amd.mm.dcn import DeformConv2dPack
    torch.manual_seed(0)
    m = DeformConv2dPack(16, 32, 3, padding=1, groups=4)
    torch.nn.init.normal_(m.conv_offset.weight, std=0.3)
    torch.nn.init.normal_(m.conv_offset.bias, std=1.5)          # offsets well beyond one pixel, some outside the image
    x = torch.randn(2, 16, 9, 11, requires_grad=True)
    off = m.conv_offset(x)
    a = m._gather_and_gemm(x, off, torch.float32)
    b = m._sample_and_contract(x, off)
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    ga = torch.autograd.grad(a.square().sum(), [x, m.weight, m.conv_offset.weight], retain_graph=True)
    gb = torch.autograd.grad(b.square().sum(), [x, m.weight, m.conv_offset.weight])
    for u, v in zip(ga, gb):
        torch.testing.assert_close(u, v, rtol=1e-3, atol=1e-3 * float(v.abs().max()))


# ---------------------------------------------------------------------------------------------
# Test-time path: box container, multi-class NMS host logic, simple_test
# ---------------------------------------------------------------------------------------------
def test_lidar_boxes_container_known_answers():
    from omnihd_amd.mm.boxes import LiDARInstance3DBoxes, xywhr2xyxyr
    raw = torch.tensor([[1.0, 2.0, 0.5, 2.0, 4.0, 1.5, 0.3, 0.1, -0.2], [0.0, 0.0, -1.0, 1.0, 1.0, 2.0, -1.0, 0.0, 0.0]])
    b = LiDARInstance3DBoxes(raw, box_dim=9)
    assert torch.equal(b.bev, raw[:, [0, 1, 3, 4, 6]]) and torch.equal(b.dims, raw[:, 3:6]) and len(b) == 2
    torch.testing.assert_close(b.gravity_center, torch.tensor([[1.0, 2.0, 1.25], [0.0, 0.0, 0.0]]))
    # dataset side (newscenes_dataset.py:273-277): annotations carry the gravity centre -> bottom centre
    c = LiDARInstance3DBoxes(raw, box_dim=9, origin=(0.5, 0.5, 0.5))
    torch.testing.assert_close(c.tensor[:, 2], torch.tensor([0.5 - 0.75, -1.0 - 1.0]))
    torch.testing.assert_close(c.gravity_center[:, 2], raw[:, 2])
    torch.testing.assert_close(xywhr2xyxyr(b.bev), torch.tensor([[0.0, 0.0, 2.0, 4.0, 0.3], [-0.5, -0.5, 0.5, 0.5, -1.0]]))
    assert len(b[1]) == 1 and len(b[torch.tensor([True, False])]) == 1 and len(LiDARInstance3DBoxes(torch.zeros(0, 9), box_dim=9)) == 0


def test_multiclass_nms_host_logic_over_the_oracle():
    from omnihd_amd.mm.boxes import LiDARInstance3DBoxes, box3d_multiclass_nms, xywhr2xyxyr
    from oracle.torch_shim import oracle_ops
    boxes = torch.tensor([[0, 0, 0, 2, 4, 1.5, 0.0], [0.2, 0, 0, 2, 4, 1.5, 0.05], [10, 0, 0, 2, 4, 1.5, 0.0],
                          [10, 0.1, 0, 2, 4, 1.5, 1.57], [30, 0, 0, 1, 1, 1, 0.0]], dtype=torch.float32)
    #                      class 0   class 1  background
    scores = torch.tensor([[0.90, 0.10, 0.0], [0.80, 0.04, 0.0], [0.30, 0.60, 0.0], [0.95, 0.50, 0.0], [0.01, 0.02, 0.0]])
    dirs = torch.tensor([0, 1, 0, 1, 0])
    cfg = dict(use_rotate_nms=True, nms_thr=0.2, score_thr=0.05, max_num=500)
    for_nms = xywhr2xyxyr(LiDARInstance3DBoxes(boxes).bev)
    with oracle_ops():
        b, s, l, d = box3d_multiclass_nms(boxes, for_nms, scores, 0.05, 500, cfg, dirs)
        # class 0: order 3,0,1,2 -> keep 3,0 (1 suppressed by 0, 2 by 3); class 1: candidates 0,2,3 -> keep 2, 0 (3 suppressed by 2)
        assert s.tolist() == pytest.approx([0.95, 0.90, 0.60, 0.10]) and l.tolist() == [0, 0, 1, 1] and d.tolist() == [1, 0, 0, 0]
        assert torch.equal(b, boxes[[3, 0, 2, 0]])
        b, s, l, d = box3d_multiclass_nms(boxes, for_nms, scores, 0.05, 3, cfg, dirs)      # global top-max_num
        assert s.tolist() == pytest.approx([0.95, 0.90, 0.60]) and l.tolist() == [0, 0, 1]
        b, s, l = box3d_multiclass_nms(boxes, for_nms, scores, 0.99, 3, cfg)               # nothing above the threshold
        assert b.shape == (0, 7) and s.shape == (0,) and l.dtype == torch.long
    with pytest.raises(NotImplementedError):
        box3d_multiclass_nms(boxes, for_nms, scores, 0.05, 3, dict(use_rotate_nms=False, nms_thr=0.2))


def test_tiny_detector_simple_test_on_cpu_with_oracle_ops():
    from omnihd_amd.harness import FusionTrainStep
    from oracle.torch_shim import oracle_ops
    torch.set_num_threads(4)
    with oracle_ops():
        st = FusionTrainStep(res="tiny", batch=2, radar_dims=7, device="cpu", dtype="fp32", channels_last=False, sets=1)
        m, b = st.raw_model, st.batches[0]
        m.eval()
        torch.nn.init.constant_(m.pts_bbox_head.conv_cls.bias, 0.0)
        out = m(return_loss=False, points=[b["points"]], img_metas=[b["img_metas"]], img=[b["img"]])
        with pytest.raises(TypeError):
            m(return_loss=False, points=b["points"][0], img_metas=[b["img_metas"]], img=[b["img"]])
        with pytest.raises(NotImplementedError):       # a list of two "augmentations"
            m(return_loss=False, points=b["points"], img_metas=b["img_metas"], img=b["img"])
    assert len(out) == 2
    for r in out:
        d = r["pts_bbox"]
        n = len(d["boxes_3d"])
        assert 0 < n <= 500 and d["scores_3d"].shape == (n,) and d["labels_3d"].shape == (n,)
        assert float(d["scores_3d"].min()) > 0.05 and set(d["labels_3d"].tolist()) <= {0, 1}
        yaw = d["boxes_3d"].yaw                      # direction fix-up: yaw in [dir_offset, dir_offset + 2pi)
        assert float(yaw.min()) >= 0.7854 - 1e-4 and float(yaw.max()) < 0.7854 + 2 * math.pi + 1e-4


def test_plan_cache_key_covers_every_transform_that_shapes_the_geometry(lss_small=None):
    """A caller that changes only post_trans / extra_rots (image or BEV augmentation) must get a fresh pooling plan; the
    same transforms again hit the cache."""
    from oracle.torch_shim import oracle_ops
    from projects.mmdet3d_plugin.bevfusion.detectors import LiftSplatShoot
    from tests.test_lss_plain_cpu import CFG
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lss_golden.npz"))
    rots, trans = torch.from_numpy(g["l1_rots"]), torch.from_numpy(g["l1_trans"])
    B, N = trans.shape[:2]
    with oracle_ops():
        net = LiftSplatShoot(**CFG)
        none = (None, None, None, None)
        p0 = net._plan_for(rots, trans, none)
        assert net._plan_for(rots, trans, none) is p0
        pt = torch.zeros(B, N, 3)
        p1 = net._plan_for(rots, trans, (None, pt, None, None))
        assert p1 is not p0 and net._plan_for(rots, trans, (None, pt.clone(), None, None)) is p1
        pt2 = pt.clone(); pt2[..., 0] = 1.5
        p2 = net._plan_for(rots, trans, (None, pt2, None, None))
        assert p2 is not p1 and p2.tabs[0] is not None and not np.array_equal(p2.tabs[1], p1.tabs[1])
        er = torch.eye(3).repeat(B, N, 1, 1)
        er[..., 0, 0] = -1.0                                        # a BEV flip handed in as extra_rots
        p3 = net._plan_for(rots, trans, (None, None, er, None))
        assert p3 is not p0 and not np.array_equal(p3.tabs[0], p0.tabs[0])
        # the same bytes in a different slot are a different geometry
        assert net._plan_for(rots, trans, (None, None, None, pt2)) is not p2


def test_radar_side_thread_is_refused_when_another_branch_also_synchronises(monkeypatch):
    """Dual-stream forward: safe on one rank always; on several ranks only while every synchronised norm layer sits in
    the radar branch (whose collectives then all come from the one side thread)."""
    import torch.distributed as dist
    from omnihd_amd.harness import tiny_model_cfg
    from omnihd_amd.mm.config import build_detector
    from omnihd_amd.mm.sync_bn import NaiveSyncBatchNorm2d
    m = build_detector(tiny_model_cfg(7))
    assert m._side_thread_is_safe()                               # no process group
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    assert m._side_thread_is_safe()                               # naiveSyncBN only inside pts_* modules
    m2 = build_detector(tiny_model_cfg(7))
    m2.reduc_conv.bn = NaiveSyncBatchNorm2d(m2.reduc_conv.bn.num_features)
    assert not m2._side_thread_is_safe()
    m3 = build_detector(tiny_model_cfg(7))
    m3.img_neck.add_module("extra_sync", torch.nn.SyncBatchNorm(8))
    assert not m3._side_thread_is_safe()


def test_camera_only_config_is_the_reference_stage1_config():
    """BASELINE.json configs[1]: harness.camera_model_cfg restates projects/configs/bevfusion_NewScenes/cam_stream/LSS.py:30-123
    (lc_fusion=False, SyncBN everywhere, head on the 256-channel camera BEV): same dict as the reference file (geometry keys
    aside) and the same state-dict keys / parameter count when built."""
    from omnihd_amd import harness
    from omnihd_amd.mm.config import build_detector, load_config
    cfg = harness.camera_model_cfg(harness.reference_model_cfg())
    assert cfg["lc_fusion"] is False and cfg["norm_cfg"]["type"] == "SyncBN" and cfg["pts_bbox_head"]["in_channels"] == 256
    assert "pts_voxel_encoder" not in cfg and cfg["img_backbone"]["norm_eval"] is False
    ref_file = "/root/reference/projects/configs/bevfusion_NewScenes/cam_stream/LSS.py"
    if os.path.exists(ref_file):
        ref = load_config(ref_file)["model"]
        assert set(ref) == set(cfg)
        for k in ref:
            if k not in ("final_dim", "img_neck", "pc_range", "pts_bbox_head"):
                assert ref[k] == cfg[k], k
        a, b = build_detector(cfg), build_detector(ref)
        assert list(a.state_dict()) == list(b.state_dict())
        assert sum(p.numel() for p in a.parameters()) == sum(p.numel() for p in b.parameters())
