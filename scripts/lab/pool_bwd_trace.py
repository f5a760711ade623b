#!/usr/bin/env python3
"""Per-workgroup phase timeline of the patch backward (trace build: -DOMNIHD_POOL_TRACE).
    OMNIHD_LIB_PATH=$PWD/scripts/micro/abl/libomnihd_trace.so python3 scripts/lab/pool_bwd_trace.py
Stamps: 0 entry, 1 patch id here, 2 loads of the depth block / pointers / feature row issued and the wave's trip count known
(pointers returned), 3 barrier (depth block in LDS), 4 wave 0 done with its points, 5 all waves done (feat_grad stored),
6 depth_grad stores issued."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import numpy as np
import torch
import bench
from omnihd_amd import ops
from omnihd_amd._lib import lib

wl = bench.BevOps("r1", 1, torch.device("cuda:0"), 1234)
plan = wl.plan


def run(s):
    wl.pool_bwd(s)


t = bench.time_kernel(run, len(wl.sets), 60)
n_blocks = int(plan.patch_order.numel())
print(f"launch mean {t*1e6:.1f} us, blocks {n_blocks}")
trace = torch.zeros(n_blocks + 64, 8, dtype=torch.int64, device="cuda:0")
L = lib()
L.omnihd_lab_set_trace.argtypes = [ctypes.c_void_p]
L.omnihd_lab_set_trace.restype = ctypes.c_int
assert L.omnihd_lab_set_trace(ctypes.c_void_p(trace.data_ptr())) == 0
for k in range(8):
    run(k % 4)
torch.cuda.synchronize()
trace.zero_()
torch.cuda.synchronize()
run(1)
torch.cuda.synchronize()
tr = trace.cpu().numpy().astype(np.float64)
ok = (tr[:, 0] > 0) & (tr[:, 6] > 0)
print("blocks with full timeline", int(ok.sum()))
us = lambda x: x / 100.0
names = ["patch id", "issue loads + pointers back", "barrier (depth in LDS)", "points (wave 0)", "other waves + feat_grad", "depth_grad stores issued"]
d = np.diff(tr[ok][:, :7], axis=1)
print("phase                                      mean    p50    p90    max   (us)")
for k, n in enumerate(names):
    x = us(d[:, k])
    print(f"{n:40s} {x.mean():6.2f} {np.median(x):6.2f} {np.percentile(x, 90):6.2f} {x.max():6.2f}")
tot = us(tr[ok, 6] - tr[ok, 0])
print(f"{'whole workgroup':40s} {tot.mean():6.2f} {np.median(tot):6.2f} {np.percentile(tot, 90):6.2f} {tot.max():6.2f}")
t0 = tr[ok, 0].min()
start, end = us(tr[ok, 0] - t0), us(tr[ok, 6] - t0)
print(f"kernel span (first entry -> last stamp) {end.max():.2f} us; starts p50 {np.median(start):.2f} p90 {np.percentile(start, 90):.2f} max {start.max():.2f}")
for tt in np.arange(0, end.max(), 4.0):
    print(f"  t={tt:5.1f} us  active workgroups {int(((start <= tt) & (end > tt)).sum())}")
