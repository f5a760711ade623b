#!/usr/bin/env python3
"""PMC calibration workload: a device copy of known size (16 B/lane reads+writes) followed by the pooling
forward, so FETCH_SIZE / WRITE_SIZE of the kernel can be corrected by the factor observed on the copy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
wl = bench.BevOps("r1", 1, torch.device("cuda:0"), 1234)
n = 128 * 1024 * 1024 // 4                                  # 128 MiB
bufs = [(torch.randn(n, device="cuda"), torch.empty(n, device="cuda")) for _ in range(4)]
for k in range(8):
    torch.mul(bufs[k % 4][0], 1.5, out=bufs[k % 4][1])          # reads 128 MiB, writes 128 MiB, 16 B per lane
    wl.pool_fwd(k % len(wl.sets))
torch.cuda.synchronize()
print("copy_bytes", n * 4, "alg_bytes", wl.fwd_algorithmic_bytes())
