#!/usr/bin/env python3
"""Pooling forward at a resolution with 1 / 2 / 4 rotating buffer sets (1 = everything the launch reads stays in the Infinity
Cache; 4 x 270 MB at R2 = cold), with and without the kept zero rows.  Run under OMNIHD_LIB_PATH for ablation builds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
from omnihd_amd import ops

res = sys.argv[1] if len(sys.argv) > 1 else "r2"
tag = os.path.basename(os.environ.get("OMNIHD_LIB_PATH", "product"))
wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
D, fhw = wl.D, wl.fH * wl.fW
nbytes = wl.fwd_algorithmic_bytes()
for keep in (0, 1):
    def run(s):
        depth, feat, og, out, dg, fg, tb = wl.sets[s]
        ops.bev_pool_v2_forward_lean(depth, feat, tb[0], tb[2], tb[8], out, D, fhw, empty_rows_kept=bool(keep))
    for nsets in (1, 2, 4):
        t = min(bench.time_kernel(run, nsets, 60) for _ in range(2))
        print(f"{tag} {res} keep_zeros={keep} sets={nsets}: {t*1e6:6.1f} us  frac {nbytes/t/8e12:.3f}")
    t = bench.time_kernel_cold(run, 4)
    print(f"{tag} {res} keep_zeros={keep} sweep-cold: {t*1e6:6.1f} us  frac {nbytes/t/8e12:.3f}")
