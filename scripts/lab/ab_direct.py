#!/usr/bin/env python3
"""A/B of the pooling forward kernels: k_pool_fwd_lean2 (LDS-staged records) vs k_pool_fwd_direct (pieces walked from global
memory), same plan, same inputs: result comparison + launch time with 1 / 4 rotating buffer sets and after a 512 MiB sweep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
from omnihd_amd import ops, plan as P

for res in (sys.argv[1:] or ["r1", "r2"]):
    wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
    D, fhw = wl.D, wl.fH * wl.fW
    nbytes = wl.fwd_algorithmic_bytes()
    dt = P.direct_tables(wl.plan)
    dts = [[t.clone() for t in dt] for _ in wl.sets]

    def lean(s, keep=False):
        depth, feat, og, out, dg, fg, tb = wl.sets[s]
        ops.bev_pool_v2_forward_lean(depth, feat, tb[0], tb[2], tb[8], out, D, fhw, empty_rows_kept=keep)

    def direct(s, keep=False):
        depth, feat, og, out, dg, fg, tb = wl.sets[s]
        ops.bev_pool_v2_forward_direct(depth, feat, dts[s][0], dts[s][1], dts[s][2], tb[2], out, D, fhw, empty_rows_kept=keep)

    depth, feat, og, out, dg, fg, tb = wl.sets[0]
    out.fill_(float("nan")); lean(0); a = out.clone()
    out.fill_(float("nan")); direct(0); b = out.clone()
    torch.cuda.synchronize()
    diff = (a - b).abs()
    print(res, "lean2 vs direct: max |diff| %.3e  rel %.3e  rows differing %d of %d  nan %d" % (
        float(diff.max()), float(diff.max() / a.abs().max()), int((diff.amax(1) > 0).sum()), a.shape[0], int(torch.isnan(b).sum())))
    for s in range(len(wl.sets)):
        wl.sets[s][3].zero_(); lean(s)          # buffers whose empty rows are zero, for the keep runs
    for keep in (False, True):
        for name, fn in (("lean2 ", lean), ("direct", direct)):
            f = lambda s: fn(s, keep)
            line = []
            for nsets in (1, 4):
                t = min(bench.time_kernel(f, nsets, 60) for _ in range(2))
                line.append("sets=%d %6.1f us (%.3f)" % (nsets, t * 1e6, nbytes / t / 8e12))
            t = bench.time_kernel_cold(f, 4)
            line.append("sweep-cold %6.1f us (%.3f)" % (t * 1e6, nbytes / t / 8e12))
            print(res, name, "keep_zeros=%d" % keep, " | ".join(line))
    # result after the keep runs must still equal the reference result
    direct(0, True); torch.cuda.synchronize()
    print(res, "direct keep run equals first direct result:", bool(torch.equal(wl.sets[0][3], b)))
