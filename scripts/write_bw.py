#!/usr/bin/env python3
"""How fast can this chip WRITE?  Fill of 157 MB (the pooling forward's output) on rotating buffers, and a copy
of 101.6 -> 101.6 MB (the same total bytes as the kernel's algorithmic traffic), HIP-event timed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
n = 614400 * 64
outs = [torch.empty(n, device="cuda") for _ in range(6)]
src = [torch.randn(n * 101 // 157, device="cuda") for _ in range(6)]
dst = [torch.empty_like(s) for s in src]
small = [torch.randn(50_000_000 // 4, device="cuda") for _ in range(6)]
acc = [torch.empty(1, device="cuda") for _ in range(6)]
for name, fn, nbytes in (("fill 157.3 MB", lambda k: outs[k].zero_(), n * 4),
                         ("copy 101+101 MB", lambda k: dst[k].copy_(src[k]), src[0].numel() * 8),
                         ("fill 157.3 MB + read 50 MB (two kernels)", lambda k: (outs[k].zero_(), torch.sum(small[k], dim=0, keepdim=True, out=acc[k])), n * 4 + 50_000_000)):
    for rep in range(2):
        t = bench.time_kernel(fn, 6, 60)
        print(f"{name:45s} {t*1e6:6.1f} us  {nbytes/t/1e12:.2f} TB/s")
