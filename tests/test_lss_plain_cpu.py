"""CPU tests of the plain Lift-Splat stream (``LiftSplatShoot``) and ``BEVF_FasterRCNN`` against
vectors captured from the reference (tests/golden/make_golden_lss.py -> lss_golden.npz).  The HIP
operators are routed to the CPU oracle here (tests only); tests/test_lss_plain_gpu.py runs the same
vectors through the HIP path."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import seeded_state

SEED = 20261001
CFG = dict(lss=False, final_dim=(32, 48), camera_depth_range=[1.0, 9.0, 1.0],
           pc_range=[-8.0, -6.0, -1.0, 8.0, 6.0, 1.0], downsample=4, grid=1.0, inputC=16, camC=8)


@pytest.fixture(scope="module")
def lss_golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "lss_golden.npz"))


def _close(a, b, tol=1e-3):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6) <= tol


def test_registry_names_and_module_paths():
    import projects.mmdet3d_plugin  # noqa: F401
    from omnihd_amd.mm import DETECTORS
    from projects.mmdet3d_plugin.bevfusion.detectors import BEVF_FasterRCNN, LiftSplatShoot
    from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2 import CamEncode, QuickCumsum  # noqa: F401
    assert DETECTORS.get("BEVF_FasterRCNN") is BEVF_FasterRCNN
    for n in ("BEVFUSION_depth", "BEVF_FasterRCNN_MTL", "RCFusion_FasterRCNN"):
        assert n in DETECTORS
    assert LiftSplatShoot.__name__ == "LiftSplatShoot"


def test_state_dict_keys_are_the_reference_ones(lss_golden):
    from projects.mmdet3d_plugin.bevfusion.detectors import LiftSplatShoot
    net = LiftSplatShoot(**CFG)
    assert sorted(net.state_dict().keys()) == lss_golden["l1_keys"].tolist()
    assert net.camencode.depthnet.kernel_size == (1, 1) and net.bevencode[1].eps == 1e-5
    with pytest.raises(NotImplementedError):
        LiftSplatShoot(**dict(CFG, lss=True))


def test_forward_and_backward_match_the_reference_over_the_oracle(lss_golden):
    from oracle.torch_shim import oracle_ops
    from projects.mmdet3d_plugin.bevfusion.detectors import LiftSplatShoot
    g = lss_golden
    with oracle_ops():
        net = seeded_state(LiftSplatShoot(**CFG), SEED)
        x, rots, trans = (torch.from_numpy(g[k]) for k in ("l1_x", "l1_rots", "l1_trans"))
        net.eval()
        with torch.no_grad():
            bev, depth = net(x, rots, trans)
            vol, _ = net.get_voxels(x, rots, trans)
        assert _close(depth, g["l1_depth"], 1e-5) and _close(vol, g["l1_volume"], 1e-5)
        assert _close(bev, g["l1_bev_eval"], 1e-4)
        net.train()
        xg = x.clone().requires_grad_()
        bev_t, _ = net(xg, rots, trans)
        (bev_t * torch.from_numpy(g["l1_w"])).sum().backward()
        assert _close(bev_t.detach(), g["l1_bev_train"], 1e-4)
        assert _close(xg.grad, g["l1_x_grad"], 1e-3)
        assert _close(net.camencode.depthnet.weight.grad, g["l1_depthnet_w_grad"], 1e-3)


def test_plain_stream_refuses_cpu_tensors_without_the_shim(lss_golden):
    from projects.mmdet3d_plugin.bevfusion.detectors import LiftSplatShoot
    g = lss_golden
    net = LiftSplatShoot(**CFG).eval()
    with pytest.raises(Exception), torch.no_grad():
        net(torch.from_numpy(g["l1_x"]), torch.from_numpy(g["l1_rots"]), torch.from_numpy(g["l1_trans"]))


def test_depth_dist_loss_matches_the_reference(lss_golden):
    from projects.mmdet3d_plugin.bevfusion.detectors import BEVF_FasterRCNN
    import types
    g = lss_golden
    shell = types.SimpleNamespace(camera_depth_range=[1.0, 9.0, 1.0])
    pred, gt = torch.from_numpy(g["l2_pred"]).requires_grad_(), torch.from_numpy(g["l2_gt"])
    for m in ("kld", "mse"):
        got = BEVF_FasterRCNN.depth_dist_loss(shell, pred, gt, loss_method=m)
        assert abs(float(got) - float(g[f"l2_{m}"])) <= 1e-5 * abs(float(g[f"l2_{m}"])), m
    got.backward()
    assert torch.isfinite(pred.grad).all()
    with pytest.raises(NotImplementedError):
        BEVF_FasterRCNN.depth_dist_loss(shell, pred, gt, loss_method="l1")


def test_cumsum_trick_and_quickcumsum_match_the_reference(lss_golden):
    from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2 import QuickCumsum, cumsum_trick
    g = lss_golden
    rows, geom, ranks = (torch.from_numpy(g[k]) for k in ("l3_rows", "l3_geom", "l3_ranks"))
    s, kept = cumsum_trick(rows, geom, ranks)
    assert torch.equal(s, torch.from_numpy(g["l3_sums"])) and torch.equal(kept, torch.from_numpy(g["l3_geom_kept"]))
    rq = rows.clone().requires_grad_()
    sq, gq = QuickCumsum.apply(rq, geom, ranks)
    (sq * torch.from_numpy(g["l3_w"])).sum().backward()
    assert torch.equal(sq.detach(), s) and torch.equal(gq, kept)
    assert torch.equal(rq.grad, torch.from_numpy(g["l3_grad"]))


def test_bevf_faster_rcnn_builds_from_the_fusion_config_and_trains_a_tiny_step():
    """The reference fusion config with ``type`` switched to BEVF_FasterRCNN (the other registry name of
    the same file family): state-dict names, the pre-computed depth target and one CPU step over the oracle."""
    from omnihd_amd.harness import FusionTrainStep, tiny_model_cfg
    from omnihd_amd.mm.config import build_detector
    from oracle.torch_shim import oracle_ops
    with oracle_ops():
        st = FusionTrainStep(res="tiny", batch=2, radar_dims=7, device="cpu", seed=3, dtype="fp32", channels_last=False,
                             sets=1)
        cfg = dict(tiny_model_cfg(7), type="BEVF_FasterRCNN")
        cfg.pop("norm_cfg", None)
        m = build_detector(cfg)
        keys = set(m.state_dict())
        assert {"lift_splat_shot_vis.camencode.depthnet.weight", "lift_splat_shot_vis.camencode.depthnet.bias",
                "lift_splat_shot_vis.bevencode.10.running_var", "reduc_conv.bn.weight"} <= keys
        assert m.reduc_conv.bn.eps == 1e-3 and m.lift_splat_shot_vis.bevencode[1].eps == 1e-5
        b = st.batches[0]
        lss = m.lift_splat_shot_vis
        B, N = b["img"].shape[:2]
        rng = np.random.default_rng(0)
        tgt = torch.from_numpy(rng.uniform(size=(B, N, lss.fH, lss.fW, lss.D)).astype(np.float32))
        mind = torch.from_numpy(rng.uniform(0.0, 12.0, size=(B, N, lss.fH, lss.fW, 1)).astype(np.float32))
        img_depth = torch.cat([mind, tgt / tgt.sum(-1, keepdim=True)], -1)
        m.train()
        losses = m(return_loss=True, points=b["points"], img_metas=b["img_metas"], gt_bboxes_3d=b["gt_bboxes_3d"],
                   gt_labels_3d=b["gt_labels_3d"], img=b["img"], img_depth=img_depth)
        assert {"loss_cls", "loss_bbox", "loss_dir", "img_depth_loss"} <= set(losses)
        total = sum(v[0] if isinstance(v, (list, tuple)) else v for v in losses.values())
        total.backward()
        assert torch.isfinite(total)
        assert m.lift_splat_shot_vis.camencode.depthnet.weight.grad.abs().sum() > 0
        with pytest.raises(TypeError):
            build_detector(dict(cfg, norm_cfg=dict(type="BN")))


def test_freeze_img_freezes_the_image_branch_and_the_camera_stream():
    """Reference :77-87: with freeze_img the image backbone, neck and the whole lift module stop training."""
    from omnihd_amd.harness import tiny_model_cfg
    from omnihd_amd.mm.config import build_detector
    for typ in ("BEVF_FasterRCNN", "BEVFUSION_depth"):
        cfg = dict(tiny_model_cfg(7), type=typ, freeze_img=True)
        if typ == "BEVF_FasterRCNN":
            cfg.pop("norm_cfg", None)
        m = build_detector(cfg)
        frozen = {n.split(".")[0] for n, p in m.named_parameters() if not p.requires_grad}
        trainable = {n.split(".")[0] for n, p in m.named_parameters() if p.requires_grad}
        assert {"img_backbone", "img_neck", "lift_splat_shot_vis"} <= frozen
        assert not ({"img_backbone", "img_neck", "lift_splat_shot_vis"} & trainable)
        assert {"pts_voxel_encoder", "pts_backbone", "pts_neck", "reduc_conv", "pts_bbox_head"} <= trainable


def test_camera_only_stage1_config_trains_and_tests_a_tiny_step():
    """projects/configs/bevfusion_NewScenes/cam_stream/LSS.py:30-123 (BASELINE configs[1]'s reference artefact): the fusion
    detector without a point stream and without the fusion conv — SyncBN everywhere, head on the 256-channel camera BEV."""
    import math
    from omnihd_amd import harness
    from omnihd_amd.mm.config import build_detector, load_config
    from oracle.torch_shim import oracle_ops
    ref = "/root/reference/projects/configs/bevfusion_NewScenes/cam_stream/LSS.py"
    if os.path.exists(ref):
        full = build_detector(load_config(ref)["model"])
        keys = set(full.state_dict())
        assert not any(k.startswith(("pts_voxel", "pts_middle", "pts_backbone", "pts_neck", "reduc_conv", "seblock")) for k in keys)
        assert full.pts_bbox_head.conv_cls.in_channels == 256 and sum(p.numel() for p in full.parameters()) == 58911173
        assert type(full.img_backbone.bn1).__name__ in ("SyncBatchNorm", "NaiveSyncBatchNorm2d", "BatchNorm2d")
    cfg = harness.tiny_model_cfg(7)
    for k in ("pts_voxel_layer", "pts_voxel_encoder", "pts_middle_encoder", "pts_backbone", "pts_neck", "se"):
        cfg.pop(k, None)
    cfg.update(lc_fusion=False, norm_cfg=dict(type="SyncBN", requires_grad=True))
    cfg["img_backbone"].update(norm_cfg=dict(type="SyncBN", requires_grad=True), norm_eval=False)
    cfg["pts_bbox_head"].update(in_channels=256, feat_channels=256)
    torch.set_num_threads(4)
    torch.manual_seed(0)
    with oracle_ops():
        m = build_detector(cfg)
        b = harness.synthetic_batch("tiny", 2, 7, "cpu", 0)
        m.train()
        opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=2e-3)
        hist = []
        for _ in range(4):
            losses = m(return_loss=True, points=None, img_metas=b["img_metas"], gt_bboxes_3d=b["gt_bboxes_3d"],
                       gt_labels_3d=b["gt_labels_3d"], img=b["img"], img_depth=b["img_depth"])
            total = sum(v[0] if isinstance(v, list) else v for v in losses.values())
            opt.zero_grad()
            total.backward()
            opt.step()
            hist.append(float(total.detach()))
        assert set(losses) == {"loss_cls", "loss_bbox", "loss_dir", "img_depth_loss"}
        assert all(math.isfinite(h) for h in hist) and hist[-1] < hist[0]
        m.eval()
        torch.nn.init.constant_(m.pts_bbox_head.conv_cls.bias, 0.0)
        out = m(return_loss=False, points=[None], img_metas=[b["img_metas"]], img=[b["img"]])
    assert len(out) == 2 and all(len(r["pts_bbox"]["boxes_3d"]) > 0 for r in out)
