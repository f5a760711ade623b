#!/usr/bin/env python3
"""MIOpen baseline for the BEV-encoder convolutions (bf16, channels-last): fwd / dgrad / wgrad times."""
import os, sys, time
import torch
import torch.nn.functional as F

shapes = [("bevenc0", 1024, 1024), ("bevenc1", 1024, 512), ("bevenc2", 512, 512), ("bevenc3", 512, 256), ("reduc", 640, 384)]
H, W, B = 160, 240, 1
dev = "cuda:0"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for name, cin, cout in shapes:
    x = torch.randn(B, cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(cout, cin, 3, 3, device=dev, dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = F.conv2d(x, w, padding=1)
    g = torch.randn_like(y)
    flops = 2 * 9 * B * H * W * cin * cout
    t_f = timeit(lambda: F.conv2d(x, w, padding=1))
    t_d = timeit(lambda: torch.autograd.grad(F.conv2d(x, w.detach(), padding=1), x, g))
    t_w = timeit(lambda: torch.autograd.grad(F.conv2d(x.detach(), w, padding=1), w, g))
    print(f"{name}: {cin}->{cout}  fwd {t_f*1e6:8.1f} us {flops/t_f/1e12:7.1f} TF/s | fwd+dgrad {t_d*1e6:8.1f} us (dgrad ~{(t_d-t_f)*1e6:8.1f} us {flops/max(t_d-t_f,1e-9)/1e12:6.1f} TF/s)"
          f" | fwd+wgrad {t_w*1e6:8.1f} us (wgrad ~{(t_w-t_f)*1e6:8.1f} us {flops/max(t_w-t_f,1e-9)/1e12:6.1f} TF/s)")

print("hand-written MFMA wgrad (omnihd_conv3x3_wgrad_bf16, incl. layout transform + split-K sum):")
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "omnihd-scenes_amd")]
from omnihd_amd import ops
for name, cin, cout in shapes:
    if cin % 128 or cout % 128:
        continue
    x = torch.randn(B, cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, cout, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    flops = 2 * 9 * B * H * W * cin * cout
    t = timeit(lambda: ops.conv3x3_wgrad(x, g))
    print(f"{name}: {cin}->{cout}  wgrad {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF/s")
