#!/usr/bin/env python3
"""In-process A/B of an environment switch that is read at call time: ab_env.py NAME A B  (alternating blocks of steps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd.harness import FusionTrainStep
name, vals = sys.argv[1], sys.argv[2:4]
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype="bf16", miopen_find=True)
for _ in range(8):
    st.step()
res = {v: [] for v in vals}
for rep in range(4):
    for v in vals:
        os.environ[name] = v
        for _ in range(3):
            st.step()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(15):
            st.step()
        torch.cuda.synchronize()
        res[v].append((time.time() - t0) / 15 * 1e3)
for v in vals:
    print(name, "=", v, " ".join(f"{t:.2f}" for t in res[v]), "ms/step")
