#!/usr/bin/env python3
"""MIOpen's fp32 convolution kernels (find mode) on the BEV-sized geometries: forward, data gradient, weight gradient,
TFLOP/s against the 157 TFLOP/s fp32 MFMA peak."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401  (seeds the MIOpen user db)
import torch
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


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
    return e0.elapsed_time(e1) / n * 1e-3


geos = [(1, 160, 240, 1024, 1024, 3), (1, 160, 240, 1024, 512, 3), (1, 160, 240, 512, 512, 3), (1, 160, 240, 512, 256, 3),
        (1, 160, 240, 640, 384, 3), (6, 64, 176, 1024, 256, 3), (6, 64, 176, 256, 256, 3), (6, 64, 176, 1280, 256, 1)]
tot = [0.0, 0.0, 0.0]
for B, H, W, cin, cout, k in geos:
    x = torch.randn(B, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    flops = 2.0 * B * H * W * cin * cout * k * k
    p = k // 2
    t = [clock(lambda: torch.nn.functional.conv2d(x, w, None, 1, p))]
    for mask in ([True, False, False], [False, True, False]):
        t.append(clock(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [p, p], [1, 1], False, [0, 0], 1, mask)))
    for i in range(3):
        tot[i] += t[i]
    print(f"{B}x{H}x{W} {cin:4d}->{cout:4d} k{k}: fwd {t[0]*1e3:7.3f} ms {flops/t[0]/1e12:5.0f} TF | dgrad {t[1]*1e3:7.3f} ms {flops/t[1]/1e12:5.0f} TF"
          f" | wgrad {t[2]*1e3:7.3f} ms {flops/t[2]/1e12:5.0f} TF", flush=True)
print(f"sum: fwd {tot[0]*1e3:.2f} ms, dgrad {tot[1]*1e3:.2f} ms, wgrad {tot[2]*1e3:.2f} ms")
