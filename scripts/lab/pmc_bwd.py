#!/usr/bin/env python3
"""PMC workload: a 128 MiB device copy (calibration of FETCH_SIZE / WRITE_SIZE) followed by the pooling backward (patch
kernel) and the pooling forward, on rotating buffer sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
from omnihd_amd import ops
res = sys.argv[1] if len(sys.argv) > 1 else "r1"
wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
plan = wl.plan
n = 128 * 1024 * 1024 // 4
bufs = [(torch.randn(n, device="cuda"), torch.empty(n, device="cuda")) for _ in range(4)]
for k in range(8):
    torch.mul(bufs[k % 4][0], 1.5, out=bufs[k % 4][1])
    wl.pool_bwd(k % len(wl.sets))           # the product path: one packed table (row | depth bin << 24)
    wl.pool_fwd(k % len(wl.sets))
torch.cuda.synchronize()
print("done")
