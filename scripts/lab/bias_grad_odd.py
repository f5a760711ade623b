#!/usr/bin/env python3
"""Bias gradient over an NHWC gradient with an ODD channel count (DepthNet's 59 depth logits): formulations compared."""
import torch


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


for shape in ((6, 59, 64, 176), (6, 18, 64, 176), (6, 123, 64, 176)):
    N, C, H, W = shape
    g = torch.randn(shape, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    rows = g.permute(0, 2, 3, 1).reshape(-1, C)
    ones = torch.ones(1, rows.shape[0], device="cuda", dtype=torch.bfloat16)
    ref = g.float().sum((0, 2, 3))
    forms = {"sum((0,2,3))": lambda: g.sum(dim=(0, 2, 3)),
             "sum((0,2,3)) fp32": lambda: g.sum(dim=(0, 2, 3), dtype=torch.float32),
             "rows.sum(0)": lambda: rows.sum(dim=0),
             "rows.float().sum(0)": lambda: rows.float().sum(dim=0),
             "ones @ rows": lambda: (ones @ rows).view(-1),
             "ones32 @ rows32": lambda: (ones.float() @ rows.float()).view(-1)}
    for name, fn in forms.items():
        err = float((fn().float() - ref).abs().max() / ref.abs().max())
        print(f"{str(shape):18s} {name:22s} {clock(fn):7.1f} us  max rel err {err:.1e}", flush=True)
