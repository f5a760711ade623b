#!/usr/bin/env python3
"""List host<->device synchronisation points of one training step (torch sync debug mode) and compare the
CPU enqueue time of a step with its wall time."""
import os, sys, time, warnings, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd.harness import FusionTrainStep

st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype="bf16", miopen_find=True)
for _ in range(4):
    st.step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10):
    st.step()
t_enq = time.time() - t0
torch.cuda.synchronize()
t_all = time.time() - t0
print(f"enqueue {t_enq/10*1e3:.1f} ms/step, wall {t_all/10*1e3:.1f} ms/step")
seen = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        fr = [f for f in traceback.extract_stack() if "/omnihd-scenes_amd/" in f.filename]
        seen[f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line}" if fr else "?"] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
st.step()
torch.cuda.set_sync_debug_mode("default")
for k, v in seen.most_common():
    print(v, k)
