#!/usr/bin/env python3
"""Per-step times of the R1 training step (events after every step): python3 scripts/lab/step_times.py [bf16|fp32] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401  (seeds the MIOpen user db)
import torch
from omnihd_amd.harness import FusionTrainStep
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype=dt, miopen_find=True)
for _ in range(8):
    st.step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    st.step()
    ev[i + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print(dt, " ".join("%.1f" % t for t in ts), "| mean %.2f median %.2f" % (sum(ts) / n, sorted(ts)[n // 2]), flush=True)
