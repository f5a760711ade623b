#!/usr/bin/env python3
"""Where is the host synchronisation inside backward?  Sync debug mode 'error' + anomaly mode: the exception
carries the forward-time stack of the node whose backward synchronised."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd.harness import FusionTrainStep
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype="bf16")
for _ in range(3):
    st.step()
torch.cuda.synchronize()
b = st.batches[0]
st.opt.zero_grad(set_to_none=True)
import contextlib
with contextlib.nullcontext():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        losses = st.model(return_loss=True, **b)
    total = sum(v if torch.is_tensor(v) else sum(v) for v in losses.values())
    torch.cuda.set_sync_debug_mode("error")
    try:
        total.backward()
        print("backward: no sync")
    except Exception as e:
        print("BACKWARD SYNC:", str(e)[:3000])
    torch.cuda.set_sync_debug_mode("error")
    try:
        torch.nn.utils.clip_grad_norm_(st.params, max_norm=35, norm_type=2)
        print("clip: no sync")
    except Exception as e:
        print("CLIP SYNC:", str(e)[:500]); traceback.print_exc(limit=6)
    try:
        st.opt.step()
        print("opt: no sync")
    except Exception as e:
        print("OPT SYNC:", str(e)[:500]); traceback.print_exc(limit=8)
    torch.cuda.set_sync_debug_mode("default")
