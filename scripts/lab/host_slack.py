#!/usr/bin/env python3
"""Is the training step bound by the host's launch rate?  Every call into the library is given an extra busy-wait; if the step
time does not move, the GPU is the bound and the host has at least that much slack.  python3 scripts/lab/host_slack.py [fp32|bf16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401
import torch
from omnihd_amd import ops
from omnihd_amd.harness import FusionTrainStep
dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype=dt, miopen_find=True)
for _ in range(5):
    st.step()
torch.cuda.synchronize()
orig = ops._raw_stream
calls = [0]
def timed(n=20):
    calls[0] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        st.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n, calls[0] / n
def host_only(n=10):
    """host time from the first launch of a step to the return of step() (the GPU still busy), steady state"""
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); st.step(); ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    return sorted(ts)[len(ts) // 2]
print(f"{dt}: step {timed()[0]:.2f} ms; host time inside step() (median, includes the wait for the voxel count) {host_only():.2f} ms")
for us in (2, 5, 10, 20):
    def slow(us=us):
        calls[0] += 1
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < us:
            pass
        return orig()
    for part in ops.PARTS:                      # (each part of the package holds its own reference)
        if hasattr(part, "_raw_stream"):
            part._raw_stream = slow
    ms, n = timed()
    print(f"  + {us:2d} us busy-wait in each of {n:.0f} library calls per step (= {us * n / 1e3:.1f} ms of host time): step {ms:.2f} ms")
for part in ops.PARTS:
    if hasattr(part, "_raw_stream"):
        part._raw_stream = orig
