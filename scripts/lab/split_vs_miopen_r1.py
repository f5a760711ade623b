#!/usr/bin/env python3
"""Deviation of the R1 fp32 forward on the split (3-term bf16) convolution kernels from the same forward on MIOpen's fp32
kernels: depth distribution, fused BEV feature, box regressions, class logits — BatchNorm in inference mode (default running
statistics: activations grow through the random-init DepthNet, its softmax runs on logits of several hundred) and in training
mode (batch statistics, the conditioning of a real step)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch  # noqa: E402

from omnihd_amd.harness import FusionTrainStep  # noqa: E402


def run(policy, train_bn):
    os.environ["OMNIHD_FP32_CONV"] = policy
    st = FusionTrainStep(res="r1", batch=1, radar_dims=7, device="cuda:0", seed=5, dtype="fp32", sets=1)
    m, b = st.raw_model, st.batches[0]
    m.train(train_bn)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    with torch.no_grad():
        fd = m.extract_feat(b["points"], img=b["img"], img_metas=b["img_metas"])
        cls, reg, dirs = m.pts_bbox_head(fd["pts_feats"])
    lss = m.lift_splat_shot_vis
    return dict(bev=fd["pts_feats"][0].float(), depth=fd["depth_dist"].float(), cls=cls[0].float(), reg=reg[0].float())


for train_bn in (False, True):
    a, b = run("split", train_bn), run("miopen", train_bn)
    print("BatchNorm", "training mode" if train_bn else "inference mode")
    for k in a:
        d = (a[k] - b[k])
        print("  %-6s max|diff|/max|ref| %.2e   rel L2 %.2e   max|ref| %.3g" % (k, float(d.abs().max() / b[k].abs().max()),
                                                                               float(d.norm() / b[k].norm()), float(b[k].abs().max())))
