#!/usr/bin/env python3
"""Host time of the weight-image refresh inside the R1 step loop (bf16 / fp32)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401
import torch
from omnihd_amd.harness import FusionTrainStep
from omnihd_amd import ops
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype=dt, miopen_find=True)
ob, os_, tb = ops.refresh_bf16_shadows, ops.refresh_split_shadows, ops._weight_image_table
acc = {"bf16": 0.0, "split": 0.0, "table_miss": 0, "n": 0}
def wrap_b():
    t = time.perf_counter(); r = ob(); acc["bf16"] += time.perf_counter() - t; return r
def wrap_s():
    t = time.perf_counter(); r = os_(); acc["split"] += time.perf_counter() - t; return r
def wrap_t(records, dev):
    n0 = len(ops._WIMG_TABLES); r = tb(records, dev); acc["table_miss"] += int(len(ops._WIMG_TABLES) != n0); return r
ops.refresh_bf16_shadows, ops.refresh_split_shadows, ops._weight_image_table = wrap_b, wrap_s, wrap_t
for _ in range(8):
    st.step()
torch.cuda.synchronize()
for k in ("bf16", "split"): acc[k] = 0.0
acc["table_miss"] = 0
t0 = time.perf_counter()
for _ in range(16):
    st.step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(dt, "per step: wall %.2f ms, host enqueue %.2f ms, refresh_bf16 %.2f ms, refresh_split %.2f ms, table misses %d" %
      (wall / 16 * 1e3, host / 16 * 1e3, acc["bf16"] / 16 * 1e3, acc["split"] / 16 * 1e3, acc["table_miss"]))
