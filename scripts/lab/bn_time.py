#!/usr/bin/env python3
"""Training-mode BatchNorm(+ReLU) kernels at the BEV sizes: time per direction against the bytes a direction must move
(forward: x twice + y once; backward: gy, y and x for the sums, gy, x (, y) again + gx)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops
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


for dt in (torch.bfloat16, torch.float32):
    for shape in ((1, 1024, 160, 240), (1, 512, 160, 240), (1, 256, 160, 240), (1, 384, 160, 240), (6, 256, 64, 176), (1, 128, 160, 240),
                  (1, 64, 160, 240)):
        c = shape[1]
        x = torch.randn(shape, device=dev).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_()
        w, b = torch.ones(c, device=dev, requires_grad=True), torch.zeros(c, device=dev, requires_grad=True)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        nbytes = x.numel() * x.element_size()
        y = ops.bn_train_act(x, w, b, rm, rv, 0.01, 1e-3, True)
        g = torch.randn_like(y)
        tf = clock(lambda: ops.bn_train_act(x, w, b, rm, rv, 0.01, 1e-3, True))
        tfb = clock(lambda: torch.autograd.grad(ops.bn_train_act(x, w, b, rm, rv, 0.01, 1e-3, True), [x, w, b], g))
        tb = tfb - tf
        print(f"{str(dt):15s} {str(shape):22s} {nbytes/1e6:6.1f} MB: fwd {tf*1e6:7.1f} us = {3*nbytes/tf/1e12:4.2f} TB/s (3 passes) | "
              f"bwd {tb*1e6:7.1f} us = {6*nbytes/tb/1e12:4.2f} TB/s (6 passes)", flush=True)
