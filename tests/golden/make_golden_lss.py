#!/usr/bin/env python3
"""Golden fixture for the plain Lift-Splat stream and the BEVF_FasterRCNN depth loss, produced by
running the REFERENCE's own Python (authoring container only; needs /root/reference):

* ``LiftSplatShoot`` (bevfusion/detectors/cam_stream_lss_bevpoolv2.py:149-373) built whole — depth
  head, BEV encoder — and run forward on a tiny 3-camera rig, eval-mode and train-mode BatchNorm;
* ``BEVF_FasterRCNN.depth_dist_loss`` (bevf_faster_rcnn.py:221-235), 'kld' and 'mse'.

Absent third-party packages are replaced by the inert stand-in modules of make_golden.py, and the
``bev_pool_v2_ext`` calls go to oracle/ (the reference has no CPU kernels).  Weights are NOT stored:
both this script and the tests fill the modules with tests/helpers.seeded_state(seed).  Only
inputs and expected outputs go into lss_golden.npz.

Usage:  python tests/golden/make_golden_lss.py
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as G  # noqa: E402
from tests.helpers import seeded_state  # noqa: E402

SEED = 20261001


def main():
    G.install_stubs()
    G.install_ext_via_oracle()
    # names bevf_faster_rcnn.py imports on top of make_golden's stand-ins (never called here)
    cnn = sys.modules["mmcv.cnn"]
    for n in ("build_upsample_layer", "constant_init", "is_norm", "kaiming_init", "xavier_init"):
        setattr(cnn, n, None)
    sys.modules["mmdet.models"].DETECTORS = G._Registry()
    G._mod("mmdet3d.models.detectors", MVXFasterRCNN=type("MVXFasterRCNN", (nn.Module,), {}))

    lss = G.load_ref("projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2",
                     "projects/mmdet3d_plugin/bevfusion/detectors/cam_stream_lss_bevpoolv2.py")
    det = G.load_ref("projects.mmdet3d_plugin.bevfusion.detectors.bevf_faster_rcnn",
                     "projects/mmdet3d_plugin/bevfusion/detectors/bevf_faster_rcnn.py")
    from oracle import lss_oracle as O

    out = {}
    # ---- L1: whole LiftSplatShoot on a tiny rig ----------------------------------------------
    cfg = dict(lss=False, final_dim=(32, 48), camera_depth_range=[1.0, 9.0, 1.0],
               pc_range=[-8.0, -6.0, -1.0, 8.0, 6.0, 1.0], downsample=4, grid=1.0, inputC=16, camC=8)
    net = seeded_state(lss.LiftSplatShoot(**cfg), SEED)
    out["l1_keys"] = np.array(sorted(net.state_dict().keys()))
    l2i = O.synthetic_rig(32, 48, 30.0, yaws_deg=(0, 90, 180), radius=0.5, height=0.3)
    rots1, trans1 = G.rig_rots_trans(l2i)
    rots, trans = torch.cat([rots1, rots1.flip(1)], 0), torch.cat([trans1, trans1.flip(1)], 0)
    x = torch.from_numpy(np.random.default_rng(5).normal(size=(2, 3, 16, 8, 12)).astype(np.float32))
    out["l1_lidar2img"], out["l1_rots"], out["l1_trans"], out["l1_x"] = l2i, rots.numpy(), trans.numpy(), x.numpy()
    net.eval()
    with torch.no_grad():
        bev, depth = net(x, rots, trans)
    out["l1_bev_eval"], out["l1_depth"] = bev.numpy(), depth.numpy()
    # the pooled volume before the BEV encoder (B, C, Z, Y, X)
    with torch.no_grad():
        vol, _ = net.get_voxels(x, rots, trans)
    out["l1_volume"] = vol.numpy()
    net.train()
    xg = x.clone().requires_grad_()
    bev_t, _ = net(xg, rots, trans)
    w = torch.from_numpy(np.random.default_rng(6).normal(size=tuple(bev_t.shape)).astype(np.float32))
    (bev_t * w).sum().backward()
    out["l1_bev_train"], out["l1_w"], out["l1_x_grad"] = bev_t.detach().numpy(), w.numpy(), xg.grad.numpy()
    out["l1_depthnet_w_grad"] = net.camencode.depthnet.weight.grad.numpy()
    print("LiftSplatShoot:", tuple(bev.shape), tuple(depth.shape), "volume", tuple(vol.shape),
          "nonzero voxels", int((vol.abs().sum(1) > 0).sum()))

    # ---- L2: depth_dist_loss -----------------------------------------------------------------
    shell = types.SimpleNamespace(camera_depth_range=[1.0, 9.0, 1.0])
    rng = np.random.default_rng(9)
    B, N, D, H, W = 2, 3, 8, 8, 12
    pred = torch.from_numpy(rng.normal(size=(B, N, D, H, W)).astype(np.float32)).softmax(2)
    tgt = torch.from_numpy(rng.uniform(size=(B, N, H, W, D)).astype(np.float32))
    tgt = tgt / tgt.sum(-1, keepdim=True)
    mind = torch.from_numpy(rng.uniform(-2.0, 12.0, size=(B, N, H, W, 1)).astype(np.float32))
    gt = torch.cat([mind, tgt], -1)
    out["l2_pred"], out["l2_gt"] = pred.numpy(), gt.numpy()
    for m in ("kld", "mse"):
        out[f"l2_{m}"] = det.BEVF_FasterRCNN.depth_dist_loss(shell, pred, gt, loss_method=m).numpy()
        print("depth_dist_loss", m, float(out[f"l2_{m}"]))

    # ---- L3: cumsum_trick / QuickCumsum --------------------------------------------------------
    ranks = torch.tensor([0, 0, 0, 2, 5, 5, 9, 9, 9, 9], dtype=torch.long)
    rows = torch.from_numpy(rng.normal(size=(10, 4)).astype(np.float32))
    geom = torch.arange(40).view(10, 4)
    s, g = lss.cumsum_trick(rows, geom, ranks)
    rq = rows.clone().requires_grad_()
    sq, gq = lss.QuickCumsum.apply(rq, geom, ranks)
    wq = torch.from_numpy(rng.normal(size=tuple(sq.shape)).astype(np.float32))
    (sq * wq).sum().backward()
    out["l3_ranks"], out["l3_rows"], out["l3_geom"] = ranks.numpy(), rows.numpy(), geom.numpy()
    out["l3_sums"], out["l3_geom_kept"], out["l3_w"], out["l3_grad"] = s.numpy(), g.numpy(), wq.numpy(), rq.grad.numpy()
    assert torch.equal(s, sq.detach()) and torch.equal(g, gq)

    path = os.path.join(HERE, "lss_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    main()
