#!/usr/bin/env python3
"""Sum over H x W of a channels-last bf16 activation (the gradient of a broadcast): formulations compared."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import torch.nn.functional as F


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


for shape in ((1, 384, 160, 240), (6, 256, 64, 176)):
    N, C, H, W = shape
    x = torch.randn(shape, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    ref = x.float().sum((2, 3), keepdim=True)
    forms = {
        "sum((2,3))": lambda: x.sum(dim=(2, 3), keepdim=True),
        "rows.sum(1)": lambda: x.permute(0, 2, 3, 1).reshape(N, H * W, C).sum(dim=1).view(N, C, 1, 1),
        "avg_pool*HW": lambda: F.adaptive_avg_pool2d(x, 1) * (H * W),
        "ones @ rows": lambda: torch.matmul(torch.ones(N, 1, H * W, device="cuda", dtype=torch.bfloat16),
                                            x.permute(0, 2, 3, 1).reshape(N, H * W, C)).view(N, C, 1, 1),
        "fp32 sum((2,3))": lambda: x.sum(dim=(2, 3), keepdim=True, dtype=torch.float32),
    }
    for name, fn in forms.items():
        y = fn()
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        print(f"{str(shape):20s} {name:16s} {clock(fn):7.1f} us   max rel err {err:.1e}", flush=True)
