#!/usr/bin/env python3
"""128x128 split kernel with a 4-stage ring (one workgroup per CU) vs a 2-stage ring (two per CU) on image-branch geometries."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops
def clock(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B, H, W, cin, cout, k in [(6, 32, 88, 128, 128, 3), (6, 16, 44, 256, 256, 3), (6, 8, 22, 512, 512, 3), (6, 32, 88, 512, 128, 1), (6, 64, 176, 64, 256, 1),
                              (6, 16, 44, 1024, 256, 1), (6, 64, 176, 256, 256, 1), (1, 80, 120, 128, 128, 3), (1, 160, 240, 64, 64, 3), (6, 64, 176, 256, 64, 1)]:
    x = torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
    xs, ws = ops.split_f32(x), ops.split_f32(w)
    t = {tile: clock(lambda: ops.conv_fwd_split(xs, ws, None, 1, tile)) for tile in (128, 129, 256)}
    fl = 2.0 * B * H * W * cin * cout * k * k
    print(f"{B}x{H}x{W} {cin}->{cout} k{k}: " + " | ".join(f"tile {tl}: {v*1e3:6.1f} us {fl/v/1e9:5.0f} TF" for tl, v in t.items()), flush=True)
