#!/usr/bin/env python3
"""torch.profiler view of one steady-state training step: per-aten-op device time, and for the copy
family (aten::copy_ / contiguous / clone / _to_copy) the input shapes + Python call site."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from torch.profiler import ProfilerActivity, profile
from omnihd_amd.harness import FusionTrainStep

st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype="bf16", miopen_find=True)
for _ in range(6):
    st.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(3):
        st.step()
    torch.cuda.synchronize()
ka = prof.key_averages()
print(ka.table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
print("=" * 120)
ks = prof.key_averages(group_by_input_shape=True, group_by_stack_n=6)
rows = [e for e in ks if e.key in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::_to_copy", "aten::cat", "aten::add_", "aten::add",
                                    "aten::clamp_min", "aten::clamp_min_", "aten::threshold_backward", "aten::mul", "aten::fill_", "aten::zero_")]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:50]:
    stack = [s for s in e.stack if "/root/repo" in s or "omnihd" in s][:3]
    print(f"{e.key:24s} {e.self_device_time_total/3/1e3:8.3f} ms/step  n={e.count/3:6.1f}  {str(e.input_shapes)[:90]:90s} {stack}")
