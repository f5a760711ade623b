#!/usr/bin/env python3
"""One geometry, N calls of the split weight gradient (PMC / kernel-trace workload).  Usage: wgrad_one.py [B H W cin cout k]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops
B, H, W, cin, cout, k = (int(v) for v in (sys.argv[1:7] + ["1", "160", "240", "1024", "1024", "3"][len(sys.argv) - 1:]))
x = torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
g = torch.randn(B, cout, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
xs, gs = ops.split_f32(x), ops.split_f32(g)
for _ in range(10):
    dw = ops.conv_wgrad_split(xs, gs, k, 1, k // 2, 1)
torch.cuda.synchronize()
print("ok", float(dw.abs().mean()))
