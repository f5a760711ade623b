#!/usr/bin/env python3
"""Module-level goldens for the dense pieces of the path that are the reference's OWN Python (not upstream):
``ASPP`` / ``_ASPPModule`` (bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:458-560), ``Cross_Modal_Fusion``
(rcfusion/detectors/BEVCross_modal_attention.py:6-43), the FPNC tail — per-level resize + 1x1 adapter, concat, 3x3
reduce (bevfusion/necks/fpnc.py:96-118) — and ``SE_Block`` (bevfusion/detectors/bevf_faster_rcnn.py:16-25), each run
with fixed weights (rebuilt from a seed by tensor name on both sides) on fixed inputs, BatchNorm in eval mode.

Stand-ins: ``mmcv.cnn.ConvModule`` = Conv2d (bias only without a norm) -> norm -> ReLU with mmcv's attribute names
(``conv``, ``bn``, ``activate``) — the documented behaviour of the absent class, it carries arithmetic; ``mmdet``'s
``FPN`` = a shell whose forward hands back the feature maps it is given (so that only FPNC's own code runs);
``build_norm_layer`` = make_golden.py's.  Usage: python tests/golden/make_golden_modules.py"""
import os
import sys

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from tests.helpers import seeded_state  # noqa: E402


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias="auto",
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True, **_):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                              bias=(norm_cfg is None) if bias == "auto" else bias)
        if norm_cfg is not None:
            self.bn = G._build_norm_layer(norm_cfg, out_channels)[1]
        if act_cfg is not None:
            assert act_cfg["type"] == "ReLU"
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x):
        x = self.conv(x)
        if hasattr(self, "bn"):
            x = self.bn(x)
        return self.activate(x) if hasattr(self, "activate") else x


class FPN(nn.Module):
    def __init__(self, in_channels=None, out_channels=None, num_outs=None, conv_cfg=None, norm_cfg=None, act_cfg=None, **_):
        super().__init__()
        self.in_channels, self.out_channels, self.num_outs = in_channels, out_channels, num_outs

    def forward(self, x):
        return tuple(x)


def save_state(out, tag, module):
    """Weights are rebuilt by name from the seed on both sides (tests/helpers.seeded_state); only the key list is kept."""
    out[f"{tag}_keys"] = np.array(sorted(module.state_dict().keys()))


def main():
    G.install_stubs()
    G.install_ext_via_oracle()
    cnn = sys.modules["mmcv.cnn"]
    cnn.ConvModule = ConvModule
    for n in ("build_upsample_layer", "constant_init", "is_norm", "kaiming_init", "xavier_init"):
        setattr(cnn, n, None)
    sys.modules["mmdet.models"].DETECTORS = G._Registry()
    sys.modules["mmdet.models"].NECKS = G._Registry()
    G._mod("mmdet.models.necks", FPN=FPN)
    G._mod("mmdet3d.models.detectors", MVXFasterRCNN=type("MVXFasterRCNN", (nn.Module,), {}))
    for name, rel in [("projects.mmdet3d_plugin.bevfusion.necks", "projects/mmdet3d_plugin/bevfusion/necks"),
                      ("projects.mmdet3d_plugin.rcfusion.detectors", "projects/mmdet3d_plugin/rcfusion/detectors")]:
        G._mod(name).__path__ = [os.path.join(G.REF, rel)]
    lss = G.load_ref("projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet",
                     "projects/mmdet3d_plugin/bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py")
    G.load_ref("projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2",
               "projects/mmdet3d_plugin/bevfusion/detectors/cam_stream_lss_bevpoolv2.py")
    det = G.load_ref("projects.mmdet3d_plugin.bevfusion.detectors.bevf_faster_rcnn",
                     "projects/mmdet3d_plugin/bevfusion/detectors/bevf_faster_rcnn.py")
    cm = G.load_ref("projects.mmdet3d_plugin.rcfusion.detectors.BEVCross_modal_attention",
                    "projects/mmdet3d_plugin/rcfusion/detectors/BEVCross_modal_attention.py")
    fp = G.load_ref("projects.mmdet3d_plugin.bevfusion.necks.fpnc", "projects/mmdet3d_plugin/bevfusion/necks/fpnc.py")
    rng = np.random.default_rng(31)
    t = lambda *s: torch.from_numpy(rng.normal(size=s).astype(np.float32))      # noqa: E731
    out = {}
    norm = dict(type="BN", eps=1e-3, momentum=0.01)

    aspp = seeded_state(lss.ASPP(16, 16, norm_cfg=norm), 1, by_name=True).eval()
    x = t(2, 16, 20, 28)                                     # dilation 18 reaches past the 20-row map
    with torch.no_grad():
        out["aspp_x"], out["aspp_y"] = x.numpy(), aspp(x).numpy()
    save_state(out, "aspp", aspp)

    cross = seeded_state(cm.Cross_Modal_Fusion(kernel_size=3, norm_cfg=norm), 2, by_name=True).eval()
    a, b = t(2, 256, 6, 8), t(2, 384, 6, 8)
    with torch.no_grad():
        out["cross_img"], out["cross_radar"], out["cross_y"] = a.numpy(), b.numpy(), cross(a, b).numpy()
    save_state(out, "cross", cross)
    cross7 = seeded_state(cm.Cross_Modal_Fusion(kernel_size=7, norm_cfg=norm), 3, by_name=True).eval()
    with torch.no_grad():
        out["cross7_y"] = cross7(a, b).numpy()
    save_state(out, "cross7", cross7)

    for tag, use_adp in (("fpnc_adp", True), ("fpnc_plain", False)):
        neck = seeded_state(fp.FPNC(final_dim=(64, 96), downsample=4, in_channels=[8, 16, 32], out_channels=8, num_outs=4,
                                    use_adp=use_adp, norm_cfg=norm if use_adp else None, act_cfg=dict(type="ReLU"), outC=12), 4, by_name=True).eval()
        levels = [t(2, 8, 32, 48), t(2, 8, 16, 24), t(2, 8, 8, 12), t(2, 8, 4, 6)]        # level 0 is LARGER than the target
        with torch.no_grad():
            y = neck(levels)
        assert isinstance(y, list) and len(y) == 1
        out[f"{tag}_y"] = y[0].numpy()
        for i, lv in enumerate(levels):
            out[f"{tag}_in{i}"] = lv.numpy()
        save_state(out, tag, neck)

    se = seeded_state(det.SE_Block(12), 5, by_name=True).eval()
    x = t(2, 12, 5, 7)
    with torch.no_grad():
        out["se_x"], out["se_y"] = x.numpy(), se(x).numpy()
    save_state(out, "se", se)

    # ---- KL depth loss of the DepthNet stream (cam_stream_lss_bevpoolv2_depthnet.py:428-456) on a shell ``self`` ----
    import types
    shell = types.SimpleNamespace(downsample=4, camera_depth_range=[1.0, 9.0, 1.0], constant_std=0.5, D=8)
    shell.get_klv_depth_loss = types.MethodType(lss.LiftSplatShoot_Depth.get_klv_depth_loss, shell)
    dm = torch.zeros(2, 3, 32, 48)
    hit = torch.from_numpy(rng.random((2, 3, 32, 48))) < 0.1
    dm[hit] = torch.from_numpy(rng.uniform(0.2, 12.0, int(hit.sum())).astype(np.float32))     # some beyond the range
    pred = torch.from_numpy(rng.normal(size=(2, 3, 8, 8, 12)).astype(np.float32)).softmax(2)
    loss, mind = lss.LiftSplatShoot_Depth.get_depth_loss(shell, dm, pred, "kld")
    out["kld_depth_map"], out["kld_pred"], out["kld_loss"], out["kld_min_depth"] = dm.numpy(), pred.numpy(), loss.numpy(), mind.numpy()
    print("kld depth loss", float(loss))

    path = os.path.join(HERE, "modules_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    main()
