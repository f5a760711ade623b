#!/usr/bin/env python3
"""Iteration timeline of k_pool_bwd_stream (lab build OMNIHD_STREAM_ABL=1, scripts/lab/patches/pool_bwd_stream_instrument.patch):
lane 0 of every wave stamps wall_clock64() (100 MHz) in its first 7 iterations at: 4t top, 4t+1 rows in LDS (= everything the
previous iteration requested has arrived), 4t+2 requests issued, 4t+3 point loop done; 30 = before the loop, 31 = wave done.
usage: OMNIHD_LIB_PATH=scripts/micro/abl/lib_pool_bwd_stream_instrument_1.so bwd_stream_trace.py r1 8 32 256"""
import ctypes, dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import numpy as np
import torch
import bench
from omnihd_amd import ops, plan as P
from omnihd_amd._lib import lib

res, pw, R, spx = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
t0 = P.stream_tables_from(wl.plan.bp_ranks_row, wl.plan.bp_ranks_depth, wl.plan.pix_ptr, wl.N, wl.D, (wl.fH, wl.fW), pw, R, spx)
tabs = [dataclasses.replace(t0, pt_word=t0.pt_word.clone(), uniq_rows=t0.uniq_rows.clone(), px_off=t0.px_off.clone(),
                            stream=t0.stream.clone(), stream_ptr=t0.stream_ptr.clone()) for _ in wl.sets]


def run(s):
    depth, feat, og, out, dg, fg, tb = wl.sets[s]
    ops.bev_pool_v2_backward_stream(og, depth, feat, tabs[s], dg, fg)


t = bench.time_kernel(run, 4, 40)
print(f"{res} pw={pw} R={R} streams/XCD={spx}: launch mean {t*1e6:.1f} us, waves {t0.n_streams}, entries {t0.stream.size(0)}")
trace = torch.zeros(t0.n_streams, 32, dtype=torch.int64, device="cuda:0")
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
tr = tr[tr[:, 31] > 0]
base = tr[:, 30].min()
us = lambda x: x / 100.0
n_it = tr[:, 29]
print("waves %d, span %.1f us; start p50 %.1f max %.1f us; lifetime p10 %.1f p50 %.1f p90 %.1f max %.1f us; iterations/wave p50 %.0f max %.0f; lifetime/iteration p50 %.2f us" % (
    len(tr), us(tr[:, 31].max() - base), *np.percentile(us(tr[:, 30] - base), [50, 100]), *np.percentile(us(tr[:, 31] - tr[:, 30]), [10, 50, 90, 100]),
    np.median(n_it), n_it.max(), np.median(us(tr[:, 31] - tr[:, 30]) / n_it)))
for it in range(7):
    m = n_it > it
    x = tr[m]
    seg = [("wait+rows->LDS", 4 * it, 4 * it + 1), ("issue", 4 * it + 1, 4 * it + 2), ("points(+stores)", 4 * it + 2, 4 * it + 3)]
    nxt = ("rotate->next top", 4 * it + 3, 4 * it + 4) if it < 6 else None
    line = " | ".join("%s p50 %.2f p90 %.2f" % (n, *np.percentile(us(x[:, b] - x[:, a]), [50, 90])) for n, a, b in seg)
    if nxt is not None:
        m2 = n_it > it + 1
        y = tr[m2]
        line += " | %s p50 %.2f" % (nxt[0], np.median(us(y[:, nxt[2]] - y[:, nxt[1]])))
    print("iteration %d (%d waves): %s" % (it, int(m.sum()), line))
