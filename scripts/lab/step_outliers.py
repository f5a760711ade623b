#!/usr/bin/env python3
"""Per-step times of the fp32 R1 step with the garbage collector's activity beside them: what makes the slow steps slow?"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401
import torch
from omnihd_amd.harness import FusionTrainStep
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype=sys.argv[1] if len(sys.argv) > 1 else "fp32", miopen_find=True)
for _ in range(5):
    st.step()
torch.cuda.synchronize()
events = []
gc.callbacks.append(lambda phase, info: events.append((time.perf_counter(), phase, info.get("generation"), info.get("collected"))))
def run(n, label):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); n0 = len(events)
        st.step(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        g = [(e[2]) for e in events[n0:] if e[1] == "start"]
        gt = 0.0
        starts = [e for e in events[n0:] if e[1] == "start"]; stops = [e for e in events[n0:] if e[1] == "stop"]
        for a, b in zip(starts, stops):
            gt += (b[0] - a[0]) * 1e3
        ts.append((dt, g, gt))
    s = sorted(t[0] for t in ts)
    print(f"{label}: median {s[len(s)//2]:.2f} mean {sum(s)/len(s):.2f} max {s[-1]:.2f} ms")
    for i, (dt, g, gt) in enumerate(ts):
        print(f"   step {i:2d} {dt:7.2f} ms  gc generations {g} gc time {gt:.2f} ms")
run(30, "gc enabled (each step synchronised)")
gc.collect(); gc.freeze(); gc.disable()
run(30, "gc frozen + disabled")
