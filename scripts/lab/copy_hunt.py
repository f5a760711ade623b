"""Where do the strided fp32 -> fp32 copies of a training step come from?  (direct_copy_kernel: 60 launches and ~1 ms per fp32 step
in profiles/round6/step_f16_first.txt.)  A few steps under torch.profiler with shapes and Python stacks; aten::copy_ / aten::add_
/ aten::cat events grouped by (shapes, innermost repo frame), sorted by device time.
Usage: python scripts/lab/copy_hunt.py [fp32|bf16] [steps]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "omnihd-scenes_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from omnihd_amd.harness import FusionTrainStep, seed_miopen_db  # noqa: E402


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    seed_miopen_db()
    st = FusionTrainStep(res="r1", batch=1, radar_dims=7, device="cuda:0", seed=1234, dtype=dt)
    for _ in range(10):
        st.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for _ in range(steps):
            st.step()
        torch.cuda.synchronize()
    groups = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.name not in ("aten::copy_", "aten::add_", "aten::add", "aten::cat", "aten::fill_", "aten::zero_", "aten::mul", "aten::sub"):
            continue
        t = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
        frame = next((s for s in ev.stack if "omnihd" in s or "projects/" in s or "harness" in s), ev.stack[0] if ev.stack else "?")
        key = (ev.name, str(ev.input_shapes)[:80], frame.strip()[-110:])
        groups[key][0] += 1
        groups[key][1] += t
    rows = sorted(groups.items(), key=lambda kv: -kv[1][1])
    print(f"{dt}: per step over {steps} steps")
    for (name, shapes, frame), (n, t) in rows[:60]:
        print(f"{t / steps:9.1f} us {n / steps:6.1f} calls  {name:12s} {shapes:80s} {frame}")


if __name__ == "__main__":
    main()
