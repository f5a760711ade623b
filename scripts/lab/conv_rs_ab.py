#!/usr/bin/env python3
"""Within-process interleaved A/B of the row-shift convolution kernel's tile codes (300 = spread A fill, 301 = all-at-once) on
the BEV geometries, bf16 and split forms: N rounds, alternating, median and minimum of the per-launch time (events around batches
of launches on random data)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops
dev = torch.device("cuda:0")
codes = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["300", "301"])]
geoms = [(1, 160, 240, 1024, 1024), (1, 160, 240, 1024, 512), (1, 160, 240, 512, 512), (1, 160, 240, 640, 384), (6, 64, 176, 1024, 256)]
rounds, per = 9, 10
for B, H, W, cin, cout in geoms:
    flops = 2.0 * B * H * W * cin * cout * 9
    xf = torch.randn(B, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    wf = (torch.randn(cout, cin, 3, 3, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    xb, wb = xf.bfloat16(), wf.bfloat16()
    xs, ws = ops.split_f32(xf), ops.split_f32(wf)
    for form in ("bf16", "split"):
        run = (lambda t: ops.conv_fwd(xb, wb, None, 1, t)) if form == "bf16" else (lambda t: ops.conv_fwd_split(xs, ws, None, 1, t))
        ref = run(codes[0])
        for c in codes[1:]:
            assert torch.equal(run(c), ref), (form, c)          # same arithmetic, same order: bit-identical results
        times = {c: [] for c in codes}
        for _ in range(3):
            for c in codes:
                run(c)
        for r in range(rounds):
            for c in (codes if r % 2 == 0 else codes[::-1]):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(per):
                    run(c)
                e1.record()
                torch.cuda.synchronize()
                times[c].append(e0.elapsed_time(e1) * 1e-3 / per)
        mult = 3 if form == "split" else 1
        row = []
        for c in codes:
            t = sorted(times[c])
            row.append(f"{c}: med {t[len(t)//2]*1e6:7.1f} us min {t[0]*1e6:7.1f} us = {flops*mult/t[len(t)//2]/1e12:6.0f} TF issued")
        print(f"{B}x{H}x{W} {cin}->{cout} {form:5s} | " + " | ".join(row), flush=True)
