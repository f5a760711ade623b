#!/usr/bin/env python3
"""Full R1 training step (bf16): per-geometry implementation choices and step time, with the convolution policy given
by OMNIHD_CONV_POLICY (tune | hip | miopen)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401  (seeds the MIOpen user db)
import torch
from omnihd_amd import ops
from omnihd_amd.harness import FusionTrainStep
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype="bf16", miopen_find=True)
for _ in range(4):
    st.step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.time()
    for _ in range(20):
        st.step()
    torch.cuda.synchronize()
    print(f"policy {os.environ.get('OMNIHD_CONV_POLICY', 'tune')}: {(time.time() - t0) / 20 * 1e3:.2f} ms/step", flush=True)
ch = ops.conv_choices()
import collections
print("conv choices:", dict(collections.Counter((k[0], v) for k, v in ch.items())))
for k, v in sorted(ch.items(), key=str):
    if v != "miopen":
        print("  ", k[0], k[1], "->", k[2], "k", k[3], v)
print("wgrad choices:", dict(collections.Counter(ops.wgrad_choices().values())))
