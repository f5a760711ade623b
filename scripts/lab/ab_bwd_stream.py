#!/usr/bin/env python3
"""A/B of the pooling backward kernels: k_pool_bwd_patch (one row gather per point) vs k_pool_bwd_stream (the distinct rows of a
patch gathered once, shared through LDS; one wave per stream of stages), same plan, same inputs: feat_grad compared bitwise,
depth_grad by its largest difference, launch time warm (1 / 4 rotating buffer sets) and after a 512 MiB sweep.
usage: ab_bwd_stream.py [r1|r2 ...] [--stream pw:R:waves_per_xcd,...]"""
import argparse, dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
from omnihd_amd import ops, plan as P

ap = argparse.ArgumentParser()
ap.add_argument("res", nargs="*", default=["r1", "r2"])
ap.add_argument("--stream", default="8:32:256,4:32:256,8:48:224,8:64:192", help="stream kernel configs 'pw:R:waves_per_xcd,...'")
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
    for cfg in (c for c in a.stream.split(",") if c):
        pw, R, spx = (int(v) for v in cfg.split(":"))
        t0 = P.stream_tables_from(wl.plan.bp_ranks_row, wl.plan.bp_ranks_depth, wl.plan.pix_ptr, wl.N, wl.D, (wl.fH, wl.fW), pw, R, spx)
        if t0 is None:
            print(res, "stream %s: tables refused" % cfg); continue
        tabs = [dataclasses.replace(t0, pt_word=t0.pt_word.clone(), uniq_rows=t0.uniq_rows.clone(), px_off=t0.px_off.clone(),
                                    stream=t0.stream.clone(), stream_ptr=t0.stream_ptr.clone()) for _ in wl.sets]

        def stream(s):
            depth, feat, og, out, dg, fg, tb = wl.sets[s]
            ops.bev_pool_v2_backward_stream(og, depth, feat, tabs[s], dg, fg)

        dg.fill_(float("nan")); fg.fill_(float("nan")); stream(0); torch.cuda.synchronize()
        same_fg = bool(torch.equal(fg, want_fg))
        dd = (dg - want_dg).abs()
        msg = "feat_grad bitwise equal %s | depth_grad max |diff| %.3e (rel to max %.3e, %d nan)" % (
            same_fg, float(dd.nan_to_num(1e30).max()), float(dd.nan_to_num(1e30).max() / want_dg.abs().max()), int(torch.isnan(dg).sum()))
        if not same_fg:
            df = (fg - want_fg).abs()
            msg += " | feat_grad max |diff| %.3e in %d rows, %d nan" % (float(df.nan_to_num(1e30).max()), int((df.nan_to_num(1e30).amax(-1) > 0).sum()), int(torch.isnan(fg).sum()))
        line = []
        for nsets in (1, 4):
            t = min(bench.time_kernel(stream, nsets, 60) for _ in range(2))
            line.append("sets=%d %6.1f us (%.3f)" % (nsets, t * 1e6, nbytes / t / 8e12))
        t = bench.time_kernel_cold(stream, 4)
        line.append("sweep-cold %6.1f us (%.3f)" % (t * 1e6, nbytes / t / 8e12))
        print(res, "stream pw=%2d R=%2d streams/XCD=%3d" % (pw, R, spx), " | ".join(line), "|", msg,
              "| reuse %.2f, entries %d, balance %.3f" % (t0.reuse, t0.stream.size(0), t0.balance), flush=True)
