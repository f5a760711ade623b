#!/usr/bin/env python3
"""Workload for rocprofv3: N launches of the pooling forward (tiled dense) and backward at R1/R2
on rotating buffer sets.  Usage: rocprofv3 ... -- python3 scripts/prof_pool.py [r1|r2] [launches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch  # noqa: E402

import bench  # noqa: E402

res = sys.argv[1] if len(sys.argv) > 1 else "r1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
for k in range(n):
    wl.pool_fwd(k % len(wl.sets))
torch.cuda.synchronize()
for k in range(n):
    wl.pool_bwd(k % len(wl.sets))
torch.cuda.synchronize()
print("done", wl.fwd_algorithmic_bytes())
