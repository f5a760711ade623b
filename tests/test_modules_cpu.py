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
