#!/usr/bin/env python3
"""GPU bring-up of the full fusion training step (not part of the bench contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd.harness import FusionTrainStep

res = sys.argv[1] if len(sys.argv) > 1 else "r1"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
t0 = time.time()
st = FusionTrainStep(res=res, batch=1, radar_dims=7 if res == "r1" else 8, dtype=dtype, miopen_find=True)
print("built", time.time() - t0)
for i in range(3):
    t0 = time.time(); loss = st.step(); torch.cuda.synchronize()
    print(i, "loss", float(loss), {k: float(v[0] if isinstance(v, list) else v) for k, v in st.last_losses.items()}, "t", time.time() - t0)
torch.cuda.synchronize(); t0 = time.time()
for i in range(10):
    st.step()
torch.cuda.synchronize(); dt = (time.time() - t0) / 10
print(f"{res} {dtype}: {dt*1e3:.1f} ms/step  {1/dt:.2f} frames/s  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
try:
    from omnihd_amd import ops
    ch = ops.wgrad_choices()
    print("wgrad choices: hip", sum(v == "hip" for v in ch.values()), "miopen", sum(v == "miopen" for v in ch.values()))
except Exception as e:
    print("no wgrad choices", e)
