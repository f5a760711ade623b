#!/usr/bin/env python3
"""A few R1 training steps (kernel-trace workload): python3 scripts/lab/step_few.py [bf16|fp32] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401  (seeds the MIOpen user db)
import torch
from omnihd_amd.harness import FusionTrainStep
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype=dt, miopen_find=os.environ.get("OMNIHD_STEP_FIND", "1") == "1")
for _ in range(3):
    st.step()
torch.cuda.synchronize()
print("MARK timed steps begin", flush=True)
for _ in range(n):
    st.step()
torch.cuda.synchronize()
print("done")
