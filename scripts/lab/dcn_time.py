#!/usr/bin/env python3
"""Deformable sampling kernels at DepthNet's size (6 x 256 x 64 x 176): forward, offset gradient + input gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float32):
    for spread in (0.0, 0.6, 2.5):
        x = torch.randn(6, 64, 176, 256, device=dev).to(dt).requires_grad_()
        off = (torch.randn(6, 64, 176, 18, device=dev) * spread).requires_grad_()
        col = ops.dcn3x3_sample(x, off, 1, 1, 1)
        g = torch.randn_like(col)

        def clock(fn, n=10):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        tf = clock(lambda: ops.dcn3x3_sample(x, off, 1, 1, 1))
        tb = clock(lambda: torch.autograd.grad(ops.dcn3x3_sample(x, off, 1, 1, 1), [x, off], g))
        print(f"{str(dt):16s} offsets sigma {spread:3.1f} (window radius {int(off.abs().max().ceil()) + 1}): fwd {tf:7.1f} us, fwd+bwd {tb:7.1f} us", flush=True)
