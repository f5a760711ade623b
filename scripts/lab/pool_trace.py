#!/usr/bin/env python3
"""Per-workgroup phase timeline of the lean pooling forward (trace build of the library, see OMNIHD_POOL_TRACE in
csrc/bev_pool_v2.hip).  Usage on the GPU box:
    OMNIHD_LIB_PATH=$PWD/scripts/micro/abl/libomnihd_trace.so python3 scripts/lab/pool_trace.py [r1]
Stamps (wall_clock64, 100 MHz): 0 entry, 1 descriptor here, 2 table loads issued + zero-fill issued, 3 records in LDS
(depth gather returned) + barrier, 4 flags + barrier, 5 wave 0 done with its points, 6 all waves done, 7 end."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import numpy as np
import torch

import bench
from omnihd_amd import ops
from omnihd_amd._lib import lib

res = sys.argv[1] if len(sys.argv) > 1 else "r1"
wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
D, fhw = wl.D, wl.fH * wl.fW
n_slots = wl.plan.tile_desc.shape[0]


def run(s):
    depth, feat, og, out, dg, fg, tb = wl.sets[s]
    ops.bev_pool_v2_forward_lean(depth, feat, tb[0], tb[2], tb[8], out, D, fhw)


t = bench.time_kernel(run, len(wl.sets), 60)
print(f"{res}: untraced-buffer launch mean {t*1e6:.1f} us, slots {n_slots}")
trace = torch.zeros(n_slots, 8, dtype=torch.int64, device="cuda:0")
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
desc = wl.plan.tile_desc.cpu().numpy()
# slot s of the schedule is processed by block b with (b & 7) * per + (b >> 3) == s; trace rows are indexed by block
per = n_slots // 8
blk = np.arange(n_slots)
slot = (blk & 7) * per + (blk >> 3)
npts = desc[slot, 3]
nrows = desc[slot, 1]
ok = (tr[:, 0] > 0) & (tr[:, 7] > 0)
print("blocks with full timeline", int(ok.sum()), "of", n_slots, "| idle slots", int((nrows <= 0).sum()),
      "| rows-only tiles", int(((npts == 0) & (nrows > 0)).sum()))
t0 = tr[tr[:, 0] > 0, 0].min()
us = lambda x: x / 100.0
names = ["desc wait", "issue L + zero-fill", "tables+depth gather -> LDS + barrier", "flags + barrier", "points (wave 0)",
         "wait other waves", "tail combine"]
d = np.diff(tr[ok], axis=1)
print("phase                                      mean    p50    p90    max   (us)")
for k, n in enumerate(names):
    x = us(d[:, k])
    print(f"{n:40s} {x.mean():6.2f} {np.median(x):6.2f} {np.percentile(x, 90):6.2f} {x.max():6.2f}")
tot = us(tr[ok, 7] - tr[ok, 0])
print(f"{'whole workgroup':40s} {tot.mean():6.2f} {np.median(tot):6.2f} {np.percentile(tot, 90):6.2f} {tot.max():6.2f}")
start = us(tr[ok, 0] - t0)
end = us(tr[ok, 7] - t0)
print(f"kernel span (first entry -> last end) {end.max():.2f} us; starts: p50 {np.median(start):.2f} p90 {np.percentile(start, 90):.2f} max {start.max():.2f}")
# residency over time
for tt in np.arange(0, end.max(), 4.0):
    act = int(((start <= tt) & (end > tt)).sum())
    print(f"  t={tt:5.1f} us  active workgroups {act}")
# correlation with work
pw = npts[ok]
for lo, hi in ((1, 200), (200, 500), (500, 700), (700, 1300), (1300, 10 ** 9)):
    m = (pw >= lo) & (pw < hi)
    if m.any():
        print(f"  tiles with {lo}-{hi} points: n={int(m.sum())} mean total {tot[m].mean():.2f} us, points phase {us(d[m, 4]).mean():.2f} us")
