#!/usr/bin/env python3
"""First measurement of BASELINE configs[4] (BEVFusionTripleTemporal, bs=2, 4-frame queue) on one GPU:
    python scripts/try_triple.py [frames=4] [batch=2] [steps=5]
Prints the wall time of every step as it finishes (the first ones carry MIOpen's per-geometry set-up for three new
batch sizes), then the per-phase GPU time of one step from torch's profiler and the peak memory.  Not run in round 1
(GPU budget); see DESIGN.md section 9, item 6."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch  # noqa: E402

from omnihd_amd.harness import FusionTrainStep  # noqa: E402

frames, batch, steps = (int(a) for a in (sys.argv[1:4] + ["4", "2", "5"][len(sys.argv) - 1:]))
t0 = time.time()
st = FusionTrainStep(res="r1", batch=batch, radar_dims=7, dtype="bf16", sets=1, task="triple", frames=frames)
print(f"built in {time.time() - t0:.1f} s; parameters {sum(p.numel() for p in st.raw_model.parameters()) / 1e6:.1f} M", flush=True)
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.time()
    loss = float(st.step().detach())
    torch.cuda.synchronize()
    print(f"step {i}: {(time.time() - t0) * 1e3:.1f} ms  loss {loss:.3f}  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB",
          flush=True)
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    st.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
