"""Layer-by-layer comparison of the plain LiftSplatShoot backward on the GPU (HIP path) against the same
weights on the CPU with the operators routed to the oracle (test infrastructure).  Prints, per stage,
max|gpu - cpu| / max|cpu| of the forward value and of the gradient, so the stage where the input-gradient
deviation of tests/test_lss_plain_gpu.py enters can be read off.  Run on the GPU box:
    python scripts/bisect_lss_grad.py [nomiopen]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "omnihd-scenes_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
from torch import nn

from oracle.torch_shim import oracle_ops
from tests.helpers import seeded_state
from tests.test_lss_plain_cpu import CFG, SEED


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def run(dev, g, use_oracle, double=False):
    import contextlib
    from projects.mmdet3d_plugin.bevfusion.detectors import LiftSplatShoot
    from omnihd_amd.mm.bricks import bn_act
    with (oracle_ops() if use_oracle else contextlib.nullcontext()):
        net = seeded_state(LiftSplatShoot(**CFG), SEED).to(dev)
        if double:
            net = net.double()
        net.train()
        cast = (lambda a: torch.from_numpy(g[a]).to(dev).double()) if double else (lambda a: torch.from_numpy(g[a]).to(dev))
        x, rots, trans, w = cast("l1_x"), cast("l1_rots"), cast("l1_trans"), cast("l1_w")
        xg = x.clone().requires_grad_()
        stages = {}

        def keep(name, t):
            t.retain_grad()
            stages[name] = t
            return t

        plan = net._plan_for(rots.float(), trans.float(), (None, None, None, None), None)
        feat, depth = net.get_cam_feats(xg)
        keep("feat", feat); keep("depth", depth)
        vol = keep("vol", net.voxel_pooling_v2(None, depth, feat, plan=plan))
        h = keep("s2c", net.s2c(vol))
        mods = list(net.bevencode)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.modules.batchnorm._BatchNorm):
                h = keep(f"bn{i}", bn_act(h, m, relu=True, inplace=False))
                i += 2
            else:
                h = keep(f"conv{i}", m(h))
                i += 1
        (h * w).sum().backward()
        out = {k: (v.detach().cpu(), v.grad.detach().cpu()) for k, v in stages.items()}
        out["x"] = (x.cpu(), xg.grad.detach().cpu())
        out["w_depthnet"] = (net.camencode.depthnet.weight.detach().cpu(), net.camencode.depthnet.weight.grad.detach().cpu())
        return out


def main():
    if "nomiopen" in sys.argv:
        torch.backends.cudnn.enabled = False
    g = np.load(os.path.join(ROOT, "tests", "golden", "lss_golden.npz"))
    cpu = run("cpu", g, True)
    cpu64 = run("cpu", g, True, double=False)
    gpu = run("cuda:0", g, False)
    print("stage            fwd(gpu vs cpu)   grad(gpu vs cpu)   | golden x_grad: gpu %.2e cpu %.2e" % (
        rel(gpu["x"][1], torch.from_numpy(g["l1_x_grad"])), rel(cpu["x"][1], torch.from_numpy(g["l1_x_grad"]))))
    for k in cpu:
        print(f"{k:14s}  {rel(gpu[k][0], cpu[k][0]):.3e}        {rel(gpu[k][1], cpu[k][1]):.3e}")
    # feed the CPU gradient of the pooled volume through the GPU pooling backward alone
    from omnihd_amd import ops  # noqa: F401
    from projects.mmdet3d_plugin.bevfusion.detectors import LiftSplatShoot
    net = seeded_state(LiftSplatShoot(**CFG), SEED).to("cuda:0").train()
    rots, trans = torch.from_numpy(g["l1_rots"]).cuda(), torch.from_numpy(g["l1_trans"]).cuda()
    plan = net._plan_for(rots, trans, (None, None, None, None), None)
    feat = cpu["feat"][0].cuda().requires_grad_()
    depth = cpu["depth"][0].cuda().requires_grad_()
    vol = net.voxel_pooling_v2(None, depth, feat, plan=plan)
    vol.backward(cpu["vol"][1].cuda())
    print("pool alone (cpu inputs, cpu out_grad): fwd %.3e  depth_grad %.3e  feat_grad %.3e" % (
        rel(vol, cpu["vol"][0]), rel(depth.grad, cpu["depth"][1]), rel(feat.grad, cpu["feat"][1])))
    # where is the x-gradient error: worst elements
    d = (gpu["x"][1] - cpu["x"][1]).abs()
    idx = torch.topk(d.flatten(), 5).indices
    for i in idx:
        print("worst x_grad elem", int(i), float(gpu["x"][1].flatten()[i]), float(cpu["x"][1].flatten()[i]))
    print("|x_grad| max", float(cpu["x"][1].abs().max()), "depth grad max", float(cpu["depth"][1].abs().max()))


if __name__ == "__main__":
    main()
