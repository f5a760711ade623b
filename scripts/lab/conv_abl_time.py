#!/usr/bin/env python3
"""Time the row-shift kernel (tile 300) on 1024->1024 @160x240, split and bf16 forms, with whatever library OMNIHD_LIB_PATH names."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops
x = torch.randn(1, 1024, 160, 240, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(1024, 1024, 3, 3, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
xs, ws = ops.split_f32(x), ops.split_f32(w)
xb, wb = x.bfloat16(), w.bfloat16().contiguous(memory_format=torch.channels_last)
def clock(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("split rs %.3f ms | bf16 rs %.3f ms" % (clock(lambda: ops.conv_fwd_split(xs, ws, None, 1, 300)), clock(lambda: ops.conv_fwd(xb, wb, None, 1, 300))))
