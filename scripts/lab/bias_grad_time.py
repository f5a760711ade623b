#!/usr/bin/env python3
"""Bias gradient of a convolution on a channels-last bf16 gradient: torch's reductions vs the column-sum kernel."""
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


for shape in ((6, 256, 64, 176), (1, 384, 160, 240), (6, 64, 64, 176)):
    g = torch.randn(shape, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    x = torch.randn(shape[0], 256, shape[2], shape[3], device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(shape[1], 256, 3, 3, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    t = {"sum((0,2,3)) bf16": clock(lambda: g.sum(dim=(0, 2, 3))),
         "sum((0,2,3)) fp32 acc": clock(lambda: g.sum(dim=(0, 2, 3), dtype=torch.float32)),
         "conv_backward bias only": clock(lambda: torch.ops.aten.convolution_backward(g, x, w, [shape[1]], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                                                      [False, False, True]))}
    if hasattr(ops, "channel_sums"):
        t["column-sum kernel"] = clock(lambda: ops.channel_sums(g))
    print(shape, {k: round(v, 1) for k, v in t.items()}, flush=True)
