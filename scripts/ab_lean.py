#!/usr/bin/env python3
"""In-run A/B of the three-table and the one-table ("lean") pooling forward kernels + result comparison."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
from omnihd_amd import ops

for res in (sys.argv[1:] or ["r1"]):
    wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
    wl.lean = False                                   # wl.pool_fwd = the three-table kernel
    D, fhw = wl.D, wl.fH * wl.fW

    def lean(s):
        depth, feat, og, out, dg, fg, tb = wl.sets[s]
        ops.bev_pool_v2_forward_lean(depth, feat, tb[0], tb[2], tb[8], out, D, fhw)

    # correctness first: same inputs, both kernels
    depth, feat, og, out, dg, fg, tb = wl.sets[0]
    wl.pool_fwd(0); a = out.clone()
    out.fill_(float("nan")); lean(0); b = out.clone()
    torch.cuda.synchronize()
    print(res, "max |diff|", float((a - b).abs().max()), "bit-identical", bool(torch.equal(a, b)), "nan", int(torch.isnan(b).sum()))
    nbytes = wl.fwd_algorithmic_bytes()
    for rep in range(3):
        t3 = bench.time_kernel(wl.pool_fwd, len(wl.sets), 60)
        t1 = bench.time_kernel(lean, len(wl.sets), 60)
        print(f"{res} rep {rep}: three-table {t3*1e6:6.1f} us ({nbytes/t3/8e12:.3f})   lean {t1*1e6:6.1f} us ({nbytes/t1/8e12:.3f})")
