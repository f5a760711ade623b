#!/usr/bin/env python3
"""Single-stream vs dual-stream (radar branch on a second thread + stream): losses of three steps and step time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd.harness import FusionTrainStep
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype="bf16", miopen_find=True)
print("dual" if os.environ.get("OMNIHD_DUAL_STREAM") == "1" else "single", [round(float(st.step()), 4) for _ in range(4)])
for _ in range(6):
    st.step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20):
    st.step()
torch.cuda.synchronize()
print(f"{(time.time()-t0)/20*1e3:.1f} ms/step")
