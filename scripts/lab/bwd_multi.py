#!/usr/bin/env python3
"""Patch backward, one patch per workgroup vs several (OMNIHD_POOL_BWD_MULTI read once per process): run this script once per
setting; the first run stores the gradients, later runs compare bit for bit.  Usage: bwd_multi.py [r1|r2] tag"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
res, tag = sys.argv[1], sys.argv[2]
wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
depth, feat, og, out, dg, fg, tb = wl.sets[0]
dg.fill_(float("nan")); fg.fill_(float("nan"))
wl.pool_bwd(0)
torch.cuda.synchronize()
ref = f"/tmp/bwd_multi_{res}.pt"
if os.path.exists(ref):
    a_dg, a_fg = torch.load(ref)
    print(tag, res, "depth_grad identical", bool(torch.equal(a_dg, dg.cpu())), "feat_grad identical", bool(torch.equal(a_fg, fg.cpu())), flush=True)
else:
    torch.save((dg.cpu(), fg.cpu()), ref)
    print(tag, res, "stored reference gradients; finite", bool(torch.isfinite(dg).all() and torch.isfinite(fg).all()), flush=True)
nb = wl.bwd_algorithmic_bytes()
ts = [bench.time_kernel(wl.pool_bwd, len(wl.sets), 60) for _ in range(4)]
print(tag, res, "MULTI=%s" % os.environ.get("OMNIHD_POOL_BWD_MULTI", "default"), " ".join(f"{t*1e6:6.1f} us ({nb/t/8e12:.3f})" for t in ts), flush=True)
