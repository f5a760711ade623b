#!/usr/bin/env python3
"""A/B of the pooling backward kernels: k_pool_bwd_patch (one row gather per point) vs k_pool_bwd_shared (distinct rows of a
patch gathered once, shared through LDS), same plan, same inputs: bitwise comparison of both gradients + launch time warm
(1 / 4 rotating buffer sets) and after a 512 MiB sweep, over patch shapes x rows per stage.
usage: ab_bwd_shared.py [r1|r2 ...] [--shapes 16,8,4] [--rows 64,96,128,160,192]"""
import argparse, dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
from omnihd_amd import ops, plan as P

ap = argparse.ArgumentParser()
ap.add_argument("res", nargs="*", default=["r1", "r2"])
ap.add_argument("--shapes", default="16,8,4")
ap.add_argument("--rows", default="64,96,128,160,192")
a = ap.parse_args()
for res in a.res:
    wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
    nbytes = wl.bwd_algorithmic_bytes()
    depth, feat, og, out, dg, fg, tb = wl.sets[0]
    dg.fill_(float("nan")); fg.fill_(float("nan")); wl.pool_bwd(0); torch.cuda.synchronize()
    want_dg, want_fg = dg.clone(), fg.clone()
    line = []
    for nsets in (1, 4):
        t = min(bench.time_kernel(wl.pool_bwd, nsets, 60) for _ in range(2))
        line.append("sets=%d %6.1f us (%.3f)" % (nsets, t * 1e6, nbytes / t / 8e12))
    t = bench.time_kernel_cold(wl.pool_bwd, 4)
    line.append("sweep-cold %6.1f us (%.3f)" % (t * 1e6, nbytes / t / 8e12))
    print(res, "patch            ", " | ".join(line), flush=True)
    for pw in (int(v) for v in a.shapes.split(",")):
        for R in (int(v) for v in a.rows.split(",")):
            t0 = P.shared_tables_from(wl.plan.bp_ranks_row, wl.plan.bp_ranks_depth, wl.plan.pix_ptr, wl.N, wl.D, (wl.fH, wl.fW), pw, R)
            if t0 is None:
                print(res, "shared pw=%d R=%d: tables refused" % (pw, R)); continue
            lds = ops.lib().omnihd_bev_pool_v2_bwd_shared_lds_bytes(R, wl.D)
            if lds > 65536:
                print(res, "shared pw=%d R=%d: %d B of LDS" % (pw, R, lds)); continue
            tabs = [dataclasses.replace(t0, pt_word=t0.pt_word.clone(), uniq_rows=t0.uniq_rows.clone(), px_stage_off=t0.px_stage_off.clone(),
                                        sched=t0.sched.clone()) for _ in wl.sets]

            def shared(s):
                depth, feat, og, out, dg, fg, tb = wl.sets[s]
                ops.bev_pool_v2_backward_shared(og, depth, feat, tabs[s], tb[11], dg, fg)

            dg.fill_(float("nan")); fg.fill_(float("nan")); shared(0); torch.cuda.synchronize()
            same = bool(torch.equal(dg, want_dg)) and bool(torch.equal(fg, want_fg))
            if not same:
                dd, df = (dg - want_dg).abs(), (fg - want_fg).abs()
                print(res, "shared pw=%d R=%d DIFFERS: depth_grad max %.3e (%d cells, %d nan), feat_grad max %.3e (%d rows, %d nan)" % (
                    pw, R, float(dd.nan_to_num(1e30).max()), int((dd > 0).sum()), int(torch.isnan(dg).sum()),
                    float(df.nan_to_num(1e30).max()), int((df.amax(-1) > 0).sum()), int(torch.isnan(fg).sum())))
            line = []
            for nsets in (1, 4):
                t = min(bench.time_kernel(shared, nsets, 60) for _ in range(2))
                line.append("sets=%d %6.1f us (%.3f)" % (nsets, t * 1e6, nbytes / t / 8e12))
            t = bench.time_kernel_cold(shared, 4)
            line.append("sweep-cold %6.1f us (%.3f)" % (t * 1e6, nbytes / t / 8e12))
            ns = t0.sched[:, 2].long()
            print(res, "shared pw=%2d R=%3d" % (pw, R), " | ".join(line), "| bitwise equal %s | reuse %.2f, stages max %d mean %.2f, LDS %d B" % (
                same, t0.reuse, t0.max_stages, float(((ns + R - 1) // R)[t0.sched[:, 0] >= 0].float().mean()), lds), flush=True)
