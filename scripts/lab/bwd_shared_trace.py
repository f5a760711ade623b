#!/usr/bin/env python3
"""Phase timeline of k_pool_bwd_shared (lab build with OMNIHD_SHARED_ABL=1, scripts/lab/patches/pool_bwd_shared_instrument.patch):
thread 0 of every workgroup stamps wall_clock64() (100 MHz) at: 0 start, 1 descriptor read, then per stage k < 3:
2+4k rows in LDS (own wave), 3+4k barrier passed, 4+4k point loop starts, 5+4k point loop done; 14 = all stages done.
usage: OMNIHD_LIB_PATH=scripts/micro/abl/lib_pool_bwd_shared_instrument_1.so bwd_shared_trace.py r1 8 64"""
import ctypes, dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import numpy as np
import torch
import bench
from omnihd_amd import ops, plan as P
from omnihd_amd._lib import lib

res, pw, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
t0 = P.shared_tables_from(wl.plan.bp_ranks_row, wl.plan.bp_ranks_depth, wl.plan.pix_ptr, wl.N, wl.D, (wl.fH, wl.fW), pw, R)
tabs = [dataclasses.replace(t0, pt_word=t0.pt_word.clone(), uniq_rows=t0.uniq_rows.clone(), px_stage_off=t0.px_stage_off.clone(),
                            sched=t0.sched.clone()) for _ in wl.sets]


def run(s):
    depth, feat, og, out, dg, fg, tb = wl.sets[s]
    ops.bev_pool_v2_backward_shared(og, depth, feat, tabs[s], tb[11], dg, fg)


t = bench.time_kernel(run, 4, 40)
n_slots = t0.sched.size(0)
print(f"{res} pw={pw} R={R}: launch mean {t*1e6:.1f} us, slots {n_slots}")
trace = torch.zeros(n_slots, 16, dtype=torch.int64, device="cuda:0")
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
live = tr[:, 14] > 0
tr = tr[live]
base = tr[:, 0].min()
us = lambda x: x / 100.0            # wall_clock64: 100 MHz
ns = (tr[:, 15].astype(np.int64) & 0xff)
print("workgroups %d, launch span %.1f us (first start .. last end)" % (len(tr), us(tr[:, 14].max() - base)))
print("start times: p50 %.1f  p90 %.1f  max %.1f us; lifetime: p10 %.1f p50 %.1f p90 %.1f us" % (
    *np.percentile(us(tr[:, 0] - base), [50, 90, 100]), *np.percentile(us(tr[:, 14] - tr[:, 0]), [10, 50, 90])))
for S in sorted(set(ns.tolist())):
    m = ns == S
    x = tr[m]
    cols = [("desc", 0, 1)]
    for k in range(min(S, 3)):
        cols += [(f"s{k} rows->LDS", 1 if k == 0 else 5 + 4 * (k - 1), 2 + 4 * k), (f"s{k} barrier", 2 + 4 * k, 3 + 4 * k),
                 (f"s{k} issue next", 3 + 4 * k, 4 + 4 * k), (f"s{k} points", 4 + 4 * k, 5 + 4 * k)]
    last = 5 + 4 * (min(S, 3) - 1) if S else 1
    cols += [("rest", last, 14)]
    print(f"stages={S}: {int(m.sum())} workgroups, lifetime p50 {np.median(us(x[:, 14] - x[:, 0])):.2f} us | " +
          " | ".join("%s %.2f" % (n, np.median(us(x[:, b] - x[:, a]))) for n, a, b in cols))
