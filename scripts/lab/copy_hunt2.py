"""Who issues the dense fp32 copies / adds / cats of a training step?  torch.profiler loses the Python stack of operators issued from
autograd's worker thread (scripts/lab/copy_hunt.py: every frame '?'); a TorchDispatchMode sees every aten call on both threads with
the Python stack of the moment — a custom Function's backward frame, or none when the C++ engine issued it itself.
Usage: python scripts/lab/copy_hunt2.py [fp32|bf16]        (OMNIHD_FP32_CONV etc. from the environment)"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "omnihd-scenes_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from omnihd_amd.harness import FusionTrainStep, seed_miopen_db  # noqa: E402

WATCH = ("copy_", "clone", "add_", "add", "cat", "contiguous", "_to_copy", "mul", "zero_", "fill_")
SEEN = collections.defaultdict(int)


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            t = next((a for a in args if isinstance(a, torch.Tensor)), None)
            if t is None and args and isinstance(args[0], (list, tuple)) and args[0]:
                t = args[0][0]
            if isinstance(t, torch.Tensor) and t.numel() >= 1 << 20:
                frames = [f for f in traceback.extract_stack()[:-1] if "omnihd" in f.filename or "mmdet3d_plugin" in f.filename]
                where = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}({f.name})" for f in frames[-3:][::-1]) or "(no repo frame: engine / torch)"
                SEEN[(name, tuple(t.shape), str(t.dtype).replace("torch.", ""), where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    seed_miopen_db()
    st = FusionTrainStep(res="r1", batch=1, radar_dims=7, device="cuda:0", seed=1234, dtype=dt)
    for _ in range(10):
        st.step()
    torch.cuda.synchronize()
    with Spy():
        st.step()
    torch.cuda.synchronize()
    print(f"{dt}: aten calls on tensors of >= 1 Mi elements in ONE step, by (op, shape, where)")
    for (name, shape, dtype, where), n in sorted(SEEN.items(), key=lambda kv: -kv[1] * (1 + len(kv[0][1]))):
        print(f"{n:4d} x {name:10s} {str(shape):28s} {dtype:8s} {where}")


if __name__ == "__main__":
    main()
