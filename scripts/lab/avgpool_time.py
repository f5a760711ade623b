#!/usr/bin/env python3
"""Global average pooling of channels-last activations: torch's adaptive_avg_pool2d((1,1)) (forward + backward) against the
column-sum kernel of csrc/batch_norm.hip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops


def clock(fn, n=20):
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


for dt in (torch.bfloat16, torch.float32):
    for shape in ((1, 384, 160, 240), (6, 256, 64, 176)):
        x = torch.randn(shape, device="cuda").to(dt).contiguous(memory_format=torch.channels_last).requires_grad_()
        pool = torch.nn.AdaptiveAvgPool2d(1)
        tf = clock(lambda: pool(x))
        y = pool(x)
        g = torch.randn_like(y)
        tb = clock(lambda: torch.autograd.grad(pool(x), x, g)) - tf
        line = f"{str(dt):15s} {str(shape):20s} torch fwd {tf:7.1f} us, bwd {tb:7.1f} us"
        if hasattr(ops, "spatial_mean"):
            t2 = clock(lambda: ops.spatial_mean(x))
            y2 = ops.spatial_mean(x)
            tb2 = clock(lambda: torch.autograd.grad(ops.spatial_mean(x), x, g)) - t2
            err = float((y2.float() - y.float()).abs().max() / y.float().abs().max())
            line += f" | ours fwd {t2:7.1f} us, bwd {tb2:7.1f} us, max rel diff {err:.1e}"
        print(line, flush=True)
