#!/usr/bin/env python3
"""Which building block breaks hipGraph capture of its backward?  Each candidate runs in its own child process
(a failed capture can take the process down): python3 scripts/lab/graph_bisect.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]

CANDS = ["plain_conv3", "bev_conv3_miopen", "bev_conv3_hip", "bev_conv1", "affine_act", "bn_train", "bn_train_torch",
         "conv_bn_relu", "relu_only", "pool", "bev_conv3_hipwgrad_only", "bev_conv3_dgrad_hip_only"]


def child(name):
    import torch
    import torch.nn as nn
    from omnihd_amd import ops
    from omnihd_amd.mm import bricks
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(2, 128, 40, 60, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    if name.startswith("real_"):
        x = torch.randn(6, 256, 64, 176, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()

    class Frozen(nn.Module):
        def __init__(self):
            super().__init__()
            self.bn = nn.BatchNorm2d(128).eval()
            for p in self.bn.parameters():
                p.requires_grad = False

        def forward(self, x):
            return bricks.bn_act(x, self.bn, relu=True, inplace=False)

    class Train(nn.Module):
        def __init__(self):
            super().__init__()
            self.bn = nn.BatchNorm2d(128)

        def forward(self, x):
            return bricks.bn_act(x, self.bn, relu=True, inplace=False)

    if name == "plain_conv3":
        mod = nn.Conv2d(128, 128, 3, padding=1, bias=False)
    elif name.startswith("bev_conv3"):
        mod = bricks.BevConv2d(128, 128, 3, padding=1, bias=False)
        os.environ["OMNIHD_CONV_POLICY"] = "miopen" if name in ("bev_conv3_miopen", "bev_conv3_hipwgrad_only") else "hip"
        os.environ["OMNIHD_WGRAD_POLICY"] = "miopen" if name in ("bev_conv3_miopen", "bev_conv3_dgrad_hip_only") else "hip"
    elif name in ("real_layer4", "real_block", "real_block_ds"):
        from omnihd_amd.mm.resnet import ResNet
        net = ResNet(depth=50, num_stages=4, out_indices=(1, 2, 3), frozen_stages=1, norm_cfg=dict(type="BN", requires_grad=False),
                     norm_eval=True, style="pytorch")
        bricks.use_bev_conv(net)
        mod = net.layer4 if name == "real_layer4" else (net.layer4[2] if name == "real_block" else net.layer4[0])
        cin = 2048 if name == "real_block" else 1024
        x = torch.randn(6, cin, 8 if name == "real_block" else 16, 22 if name == "real_block" else 44, device=dev) \
            .contiguous(memory_format=torch.channels_last).requires_grad_()
    elif name.startswith("real_conv1"):
        mod = bricks.BevConv2d(256, 128, 1, bias=False)
        os.environ["OMNIHD_WGRAD_POLICY"] = "hip" if name.endswith("hip") else "miopen"
    elif name.startswith("real_chain"):
        mod = nn.Sequential(bricks.BevConv2d(256, 128, 1, bias=False), bricks.BevConv2d(128, 128, 3, padding=1, bias=False),
                            bricks.BevConv2d(128, 512, 1, bias=False))
        os.environ["OMNIHD_WGRAD_POLICY"] = "hip" if name.endswith("hip") else "miopen"
    elif name == "bev_conv1":
        mod = bricks.BevConv2d(128, 256, 1, bias=False)
    elif name == "affine_act":
        mod = Frozen()
    elif name == "bn_train":
        mod = Train()
    elif name == "bn_train_torch":
        mod = nn.BatchNorm2d(128)
    elif name == "conv_bn_relu":
        mod = nn.Sequential(bricks.BevConv2d(128, 128, 3, padding=1, bias=False), Train())
    elif name == "relu_only":
        mod = nn.Sequential(nn.ReLU())
    elif name == "pool":
        mod = None
    else:
        raise SystemExit("unknown " + name)
    if mod is None:
        print("skipped")
        return
    mod = mod.to(dev).to(memory_format=torch.channels_last)
    mod.train()
    if name == "affine_act":
        mod.bn.eval()
    if name in ("real_layer4", "real_block", "real_block_ds"):
        for mm in mod.modules():
            if isinstance(mm, nn.modules.batchnorm._BatchNorm):
                mm.eval()
                for p_ in mm.parameters():
                    p_.requires_grad = False
    with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
        for _ in range(3):                       # eager warm-up (kernel choices, workspaces)
            y = mod(x)
            y.float().square().mean().backward()
        torch.cuda.synchronize()
        ref = mod(x)
        (gref,) = torch.autograd.grad(ref.float().square().mean(), x)
        ref = ref.detach()
        del y                                     # no autograd graph of the eager passes may stay alive: their AccumulateGrad
        for p_ in mod.parameters():               # nodes are tied to the default stream and would pull it into the capture
            p_.grad = None
        x.grad = None
        import gc; gc.collect()
        torch.cuda.synchronize()
        import copy
        eager = copy.deepcopy(mod)
        g = torch.cuda.make_graphed_callables(mod, (x.detach().clone().requires_grad_(),), num_warmup_iters=3)
        torch.cuda.synchronize()
        print("captured", flush=True)
        for trial in range(4):                    # NEW inputs: a kernel that escaped the capture would leave stale results
            x2 = (torch.randn_like(x) * (trial + 1.5)).contiguous(memory_format=torch.channels_last).requires_grad_()
            out = g(x2)
            ga = torch.autograd.grad(out.float().square().mean(), [x2] + [p_ for p_ in mod.parameters() if p_.requires_grad])
            ref = eager(x2)
            gb = torch.autograd.grad(ref.float().square().mean(), [x2] + [p_ for p_ in eager.parameters() if p_.requires_grad])
            torch.cuda.synchronize()
            rel = lambda u, v: float((u.float() - v.float()).abs().max() / v.float().abs().max().clamp_min(1e-30))
            print("trial %d: out diff %.3e  grads diff %s (rel to max)  |grad|max %s" % (
                trial, rel(out, ref), ["%.2e" % rel(u, v) for u, v in zip(ga, gb)], ["%.2e" % float(u.abs().max()) for u in ga]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        for c in (sys.argv[1:] or CANDS):
            r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), "child", c], capture_output=True, text=True)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if "amdgpu" not in l and l.strip()]
            print(f"=== {c}: rc={r.returncode}")
            for l in tail[-8:]:
                print("   ", l[:220])
