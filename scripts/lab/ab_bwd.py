#!/usr/bin/env python3
"""Pooling backward: patch kernel vs scheduled kernel (+ its memset), same inputs, timing on rotating buffer sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
from omnihd_amd import ops

for res in (sys.argv[1:] or ["r1"]):
    wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
    plan = wl.plan
    pix = [plan.pix_ptr.clone() for _ in wl.sets]
    po = [plan.patch_order.clone() for _ in wl.sets]

    def patch(s):
        depth, feat, og, out, dg, fg, tb = wl.sets[s]
        ops.bev_pool_v2_backward_patch(og, depth, feat, tb[3], tb[5], pix[s], po[s], dg, fg)

    depth, feat, og, out, dg, fg, tb = wl.sets[0]
    wl.pool_bwd(0); a_dg, a_fg = dg.clone(), fg.clone()
    dg.fill_(float("nan")); fg.fill_(float("nan")); patch(0)
    torch.cuda.synchronize()
    print(res, "feat_grad bit-identical", bool(torch.equal(a_fg, fg)), "| depth_grad max rel",
          float((a_dg - dg).abs().max() / a_dg.abs().max()), "zero pattern same", bool(torch.equal(a_dg == 0, dg == 0)), flush=True)
    nb = 177e6 if res == "r1" else 320e6
    for rep in range(3):
        t0 = bench.time_kernel(wl.pool_bwd, len(wl.sets), 60)
        t1 = bench.time_kernel(patch, len(wl.sets), 60)
        print(f"{res} rep {rep}: sched+memset {t0*1e6:6.1f} us ({nb/t0/8e12:.3f})   patch {t1*1e6:6.1f} us ({nb/t1/8e12:.3f})", flush=True)
