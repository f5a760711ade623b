#!/usr/bin/env python3
"""Which Python call sites issue the small ATen kernels of one training step?  A TorchDispatchMode counts every ATen op
of one steady-state step (forward on this thread; the backward's ops are attributed to 'autograd') with the innermost
frame of this repository that called it.  python3 scripts/lab/op_census.py [bf16|fp32]"""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from omnihd_amd.harness import FusionTrainStep
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
os.environ["OMNIHD_DUAL_STREAM"] = "0"            # one thread, so that every op is seen by the mode
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype=dt, miopen_find=False)
for _ in range(4):
    st.step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
counts = collections.Counter()
reductions = []
SKIP = ("aten.view", "aten.permute", "aten.detach", "aten.t.", "aten.expand", "aten.reshape", "aten._unsafe_view", "aten.alias",
        "aten.slice", "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.transpose", "aten.as_strided", "aten.unbind",
        "aten.split", "aten.sym_", "aten.is_", "aten.stride", "aten.size", "aten.empty", "aten.new_empty", "aten.lift_fresh",
        "aten.unflatten", "aten.flatten", "aten.chunk", "aten._local_scalar_dense")


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if ROOT in fr.filename and "op_census" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                    break
            counts[(name, site)] += 1
            if any(k in name for k in ("aten.sum", "aten.mean", "aten.linalg_vector_norm", "aten.norm", "aten.amax", "aten.max")):
                shapes = [(tuple(a.shape), str(a.dtype).replace("torch.", ""), a.stride()) for a in args if torch.is_tensor(a)]
                reductions.append((name, site, shapes, [a for a in args[1:] if not torch.is_tensor(a)]))
        return func(*args, **(kwargs or {}))


with Census():
    st.step()
torch.cuda.synchronize()
by_op = collections.Counter()
for (name, site), c in counts.items():
    by_op[name] += c
print("ATen ops in one step (views excluded):", sum(by_op.values()))
for name, c in by_op.most_common(28):
    print(f"{c:5d}  {name}")
    for (n2, site), c2 in sorted(((k, v) for k, v in counts.items() if k[0] == name), key=lambda kv: -kv[1])[:6]:
        print(f"        {c2:4d}  {site}")

print("--- reductions with big inputs")
for name, site, shapes, rest in reductions:
    if shapes and max((torch.tensor(sh[0]).prod().item() if sh[0] else 1) for sh in shapes) >= 1 << 20:
        print(name, site, shapes, rest)
