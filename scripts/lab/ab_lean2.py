#!/usr/bin/env python3
"""Lean forward kernels, correctness against the three-table kernel + timing (kernel generation chosen by OMNIHD_POOL_LEAN2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
from omnihd_amd import ops

tag = "lean2" if os.environ.get("OMNIHD_POOL_LEAN2", "1") != "0" else "lean1"
for res in (sys.argv[1:] or ["r1"]):
    wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
    wl.lean = False
    D, fhw = wl.D, wl.fH * wl.fW

    def lean(s):
        depth, feat, og, out, dg, fg, tb = wl.sets[s]
        ops.bev_pool_v2_forward_lean(depth, feat, tb[0], tb[2], tb[8], out, D, fhw)

    depth, feat, og, out, dg, fg, tb = wl.sets[0]
    wl.pool_fwd(0); a = out.clone()
    out.fill_(float("nan")); lean(0); b = out.clone()
    torch.cuda.synchronize()
    rel = float((a - b).abs().max() / a.abs().max())
    print(res, tag, "max rel diff vs three-table", rel, "bit-identical", bool(torch.equal(a, b)), "nan", int(torch.isnan(b).sum()),
          "rows differing", int((a != b).any(1).sum()), flush=True)
    nbytes = wl.fwd_algorithmic_bytes()
    for rep in range(3):
        t1 = bench.time_kernel(lean, len(wl.sets), 60)
        print(f"{res} {tag} rep {rep}: {t1*1e6:6.1f} us ({nbytes/t1/8e12:.3f})", flush=True)
