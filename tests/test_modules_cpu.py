"""The dense pieces of the path that are the reference's own Python, against outputs of the reference classes
(tests/golden/make_golden_modules.py): ASPP, Cross_Modal_Fusion, the FPNC tail (resize + adapters + concat + reduce)
and SE_Block — same weights (rebuilt by tensor name from a seed), same inputs, BatchNorm in eval mode, fp32 on the CPU
(these modules are plain torch; on the GPU their convolutions run through MIOpen)."""
import os
from unittest import mock

import numpy as np
import pytest
import torch

from tests.helpers import seeded_state

NORM = dict(type="BN", eps=1e-3, momentum=0.01)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "modules_golden.npz"))


def _close(a, b, tol):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-6)


def test_aspp_matches_the_reference(gold):
    from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet import ASPP
    m = seeded_state(ASPP(16, 16, norm_cfg=NORM), 1, by_name=True).eval()
    assert sorted(m.state_dict().keys()) == gold["aspp_keys"].tolist()
    with torch.no_grad():
        y = m(torch.from_numpy(gold["aspp_x"]))
    assert y.shape == gold["aspp_y"].shape and _close(y, gold["aspp_y"], 1e-5)


@pytest.mark.parametrize("k,tag,seed", [(3, "cross", 2), (7, "cross7", 3)])
def test_cross_modal_fusion_matches_the_reference(gold, k, tag, seed):
    from projects.mmdet3d_plugin.rcfusion.detectors.BEVCross_modal_attention import Cross_Modal_Fusion
    m = seeded_state(Cross_Modal_Fusion(kernel_size=k, norm_cfg=NORM), seed, by_name=True).eval()
    assert sorted(m.state_dict().keys()) == gold[f"{tag}_keys"].tolist()
    with torch.no_grad():
        y = m(torch.from_numpy(gold["cross_img"]), torch.from_numpy(gold["cross_radar"]))
    assert _close(y, gold[f"{tag}_y"], 1e-5)
    with pytest.raises(AssertionError):
        Cross_Modal_Fusion(kernel_size=5, norm_cfg=NORM)


@pytest.mark.parametrize("tag,use_adp", [("fpnc_adp", True), ("fpnc_plain", False)])
def test_fpnc_tail_matches_the_reference(gold, tag, use_adp):
    """FPNC's own code (everything after the upstream FPN): the FPN forward is bypassed so that the given pyramid
    reaches the tail, exactly as in the golden run."""
    from omnihd_amd.mm.fpn import FPN
    from projects.mmdet3d_plugin.bevfusion.necks.fpnc import FPNC
    m = FPNC(final_dim=(64, 96), downsample=4, in_channels=[8, 16, 32], out_channels=8, num_outs=4, use_adp=use_adp,
             norm_cfg=NORM if use_adp else None, act_cfg=dict(type="ReLU"), outC=12)
    tail = [k for k in m.state_dict() if k.startswith(("adp.", "reduc_conv."))]
    assert sorted(tail) == gold[f"{tag}_keys"].tolist()           # the reference-side shell FPN has no parameters
    seeded_state(m, 4, by_name=True).eval()
    levels = [torch.from_numpy(gold[f"{tag}_in{i}"]) for i in range(4)]
    with mock.patch.object(FPN, "forward", lambda self, x: tuple(x)), torch.no_grad():
        y = m(levels)
    assert isinstance(y, list) and len(y) == 1 and y[0].shape == (2, 12, 16, 24)
    assert _close(y[0], gold[f"{tag}_y"], 1e-5)


def test_se_block_matches_the_reference(gold):
    from projects.mmdet3d_plugin.bevfusion.detectors import SE_Block
    m = seeded_state(SE_Block(12), 5, by_name=True).eval()
    assert sorted(m.state_dict().keys()) == gold["se_keys"].tolist()
    with torch.no_grad():
        assert _close(m(torch.from_numpy(gold["se_x"])), gold["se_y"], 1e-6)


def test_kl_depth_loss_matches_the_reference_method(gold):
    """``get_depth_loss(..., 'kld')`` of the DepthNet stream: masked mean instead of the reference's boolean gather (no
    host synchronisation), same value and the same returned min-depth map; gradient finite."""
    from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet import LiftSplatShoot_Depth
    lss = LiftSplatShoot_Depth(final_dim=(32, 48), camera_depth_range=[1.0, 9.0, 1.0], pc_range=[-8.0, -6.0, -1.0, 8.0, 6.0, 1.0],
                               downsample=4, grid=1.0, inputC=16, camC=8, norm_cfg=NORM)
    pred = torch.from_numpy(gold["kld_pred"]).requires_grad_()
    loss, mind = lss.get_depth_loss(torch.from_numpy(gold["kld_depth_map"]), pred, "kld")
    assert abs(float(loss) - float(gold["kld_loss"])) <= 1e-5 * abs(float(gold["kld_loss"]))
    assert np.array_equal(mind.numpy(), gold["kld_min_depth"])
    loss.backward()
    assert torch.isfinite(pred.grad).all() and float(pred.grad.abs().sum()) > 0
    with pytest.raises(NotImplementedError):
        lss.get_depth_loss(torch.from_numpy(gold["kld_depth_map"]), pred, "bce")


@pytest.mark.parametrize("tag,cw,sin", [("cw_sin", [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2], True), ("plain", None, False)])
def test_head_loss_from_targets_matches_the_vendored_head(tag, cw, sin):
    """``Anchor3DHeadV1.loss_single`` (the copy of the anchor head the reference vendors) on synthetic maps and targets
    (tests/golden/make_golden_head.py) against ``Anchor3DHead.loss_from_targets`` with the restated upstream losses:
    the gather-free formulation (all anchors, zero weight off the positives) gives the same three numbers."""
    from omnihd_amd.mm.anchor_head import Anchor3DHead, CrossEntropyLoss, FocalLoss, SmoothL1Loss
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "head_golden.npz"))
    head = Anchor3DHead.__new__(Anchor3DHead)                      # the loss tail reads these attributes only
    torch.nn.Module.__init__(head)
    head.num_classes, head.box_code_size, head.diff_rad_by_sin = 3, 9, sin
    head.train_cfg = dict(code_weight=cw)
    head.loss_cls = FocalLoss(use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0)
    head.loss_bbox = SmoothL1Loss(beta=1.0 / 9.0, loss_weight=1.0)
    head.loss_dir = CrossEntropyLoss(use_sigmoid=False, loss_weight=0.2)
    t = lambda k: torch.from_numpy(g[k])      # noqa: E731
    out = head.loss_from_targets(t("cls"), t("box"), t("dirs"), t("labels"), t("label_w"), t("box_t"), t("box_w"),
                                 t("dir_t"), t("dir_w"), torch.tensor(float(g["num_total"])))
    got = np.array([float(out["loss_cls"][0]), float(out["loss_bbox"][0]), float(out["loss_dir"][0])])
    assert np.allclose(got, g[f"{tag}_losses"], rtol=2e-6, atol=0), (got, g[f"{tag}_losses"])
    s1, s2 = Anchor3DHead.add_sin_difference(t("sin_b1"), t("sin_b2"))
    assert np.array_equal(s1.numpy(), g["sin_o1"]) and np.array_equal(s2.numpy(), g["sin_o2"])


def test_head_test_time_branch_matches_the_vendored_head():
    """``Anchor3DHeadV1.get_bboxes_single``: what goes INTO the multi-class NMS (decoded boxes, corner form, padded score
    matrix, direction bins, thresholds) and the yaw fix-up applied to what comes OUT, with the NMS itself replaced by the
    same recording stand-in on both sides (two levels: ``nms_pre`` cuts the first only)."""
    from omnihd_amd.mm import boxes as B
    from omnihd_amd.mm.anchor_head import Anchor3DHead, DeltaXYZWLHRBBoxCoder
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "head_golden.npz"))
    head = Anchor3DHead.__new__(Anchor3DHead)
    torch.nn.Module.__init__(head)
    head.num_classes, head.box_code_size, head.use_sigmoid_cls = 3, 9, True
    head.bbox_coder, head.dir_offset, head.dir_limit_offset = DeltaXYZWLHRBBoxCoder(code_size=9), 0.7854, 0
    head.test_cfg = dict(nms_pre=40, score_thr=0.05, max_num=20, use_rotate_nms=True, nms_thr=0.2)
    seen = {}

    def nms_recorder(bboxes, for_nms, scores, score_thr, max_num, cfg, dir_scores):
        seen.update(bboxes=bboxes.clone(), for_nms=for_nms.clone(), scores=scores.clone(), thr_max=[score_thr, max_num],
                    dir_scores=dir_scores.clone())
        keep = torch.arange(0, bboxes.shape[0], 3)[:max_num]
        best, lab = scores[keep, :-1].max(dim=1)
        return bboxes[keep], best, lab, dir_scores[keep]
    t = lambda k: torch.from_numpy(g[k])      # noqa: E731
    with mock.patch.object(B, "box3d_multiclass_nms", nms_recorder):
        boxes, scores, labels = head.get_bboxes_single([t("tt_cls0"), t("tt_cls1")], [t("tt_box0"), t("tt_box1")],
                                                       [t("tt_dir0"), t("tt_dir1")], [t("tt_anc0"), t("tt_anc1")], dict())
    assert torch.allclose(seen["bboxes"], t("tt_nms_in_bboxes"), rtol=1e-6, atol=1e-6)
    assert torch.allclose(seen["for_nms"], t("tt_nms_in_for_nms"), rtol=1e-6, atol=1e-6)
    assert torch.allclose(seen["scores"], t("tt_nms_in_scores"), rtol=1e-6, atol=1e-7) and seen["scores"].shape[1] == 4
    assert torch.equal(seen["dir_scores"], t("tt_nms_in_dir_scores")) and seen["thr_max"] == g["tt_nms_in_thr_max"].tolist()
    assert torch.allclose(boxes.tensor, t("tt_boxes"), rtol=1e-6, atol=1e-6)
    assert torch.allclose(scores, t("tt_scores"), rtol=1e-6, atol=1e-7) and torch.equal(labels, t("tt_labels"))


def test_lazy_batch_counter_reaches_state_dict_and_resets_on_load():
    """bricks._count_batch keeps `num_batches_tracked` increments on the host (no kernel per layer and step on the fused
    GPU path); a state dict carries the value torch's own BatchNorm would have stored, loading one drops pending counts,
    deep copies keep their own counters."""
    import copy
    from omnihd_amd.mm import bricks
    bn = torch.nn.BatchNorm2d(8)
    for _ in range(3):
        bricks._count_batch(bn)
    assert int(bn.num_batches_tracked) == 0 and bn._omnihd_pending_batches == 3
    twin = copy.deepcopy(bn)
    bricks._count_batch(twin)
    assert int(bn.state_dict()["num_batches_tracked"]) == 3 and bn._omnihd_pending_batches == 0
    assert int(twin.state_dict()["num_batches_tracked"]) == 4
    bricks._count_batch(bn)
    bn.load_state_dict(twin.state_dict())
    assert bn._omnihd_pending_batches == 0 and int(bn.state_dict()["num_batches_tracked"]) == 4
    # the plain (torch) branch of bn_act flushes before torch's own forward reads the counter
    bricks._count_batch(bn)
    bn.train()
    bricks.bn_act(torch.randn(2, 8, 3, 3), bn, relu=False)
    assert int(bn.num_batches_tracked) == 6          # 4 + 1 pending + torch's own increment


def test_aspp_broadcast_branch_has_the_gradient_of_a_plain_expand(monkeypatch):
    """ASPP's image-pooling branch is broadcast through a custom function (packed reduction in its backward on the GPU);
    its gradients equal those of `pooled.expand(...)`."""
    import projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet as mod
    torch.manual_seed(0)
    m = mod.ASPP(16, 16, norm_cfg=NORM).train()
    m.dropout.p = 0.0
    x = torch.randn(2, 16, 7, 9, requires_grad=True)
    g = torch.randn(2, 16, 7, 9)
    params = [p for p in m.parameters()]
    y1 = m(x)
    got = torch.autograd.grad(y1, [x] + params, g)

    class Plain:
        @staticmethod
        def apply(pooled, h, w):
            return pooled.expand(-1, -1, h, w)
    monkeypatch.setattr(mod, "_BroadcastHW", Plain)
    y2 = m(x)
    want = torch.autograd.grad(y2, [x] + params, g)
    assert torch.equal(y1, y2)
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)
