#!/usr/bin/env python3
"""Which Python lines cause dtype / layout copies in one training step?  A TorchDispatchMode logs every
aten copy-like op with shapes, dtypes, strides and the innermost repo frame; backward runs on the calling
thread so it is seen too.  Usage: trace_copies.py [tiny-cpu]"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from omnihd_amd.harness import FusionTrainStep

WATCH = {"copy_", "_to_copy", "clone", "contiguous", "cat", "add_", "add", "zeros_like", "zero_", "fill_", "empty_like"}
log = collections.Counter()
bytes_ = collections.Counter()


def fmt(t):
    if not isinstance(t, torch.Tensor):
        return str(t)[:20]
    cl = t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()
    lay = "CL" if cl else ("C" if t.is_contiguous() else "strided")
    return f"{tuple(t.shape)}:{str(t.dtype)[6:]}:{lay}"


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in WATCH:
            frames = [f for f in traceback.extract_stack() if "/omnihd-scenes_amd/" in f.filename and "trace_copies" not in f.filename]
            where = f"{os.path.basename(frames[-1].filename)}:{frames[-1].lineno}" if frames else "(autograd/other)"
            targs = [a for a in args if isinstance(a, torch.Tensor)]
            if name == "cat" and args and isinstance(args[0], (list, tuple)):
                targs = list(args[0])
            key = (name, " ".join(fmt(a) for a in targs[:3]), "-> " + fmt(out) if isinstance(out, torch.Tensor) else "", where)
            log[key] += 1
            if isinstance(out, torch.Tensor):
                bytes_[key] += out.numel() * out.element_size()
        return out


tiny = len(sys.argv) > 1 and sys.argv[1] == "tiny-cpu"
torch.autograd.set_multithreading_enabled(False)
if tiny:
    from oracle.torch_shim import oracle_ops
    ctx = oracle_ops()
    ctx.__enter__()
    st = FusionTrainStep(res="tiny", batch=1, radar_dims=7, device="cpu", dtype="fp32", channels_last=False, sets=1)
else:
    st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype=(sys.argv[1] if len(sys.argv) > 1 else "bf16"), miopen_find=True)
for _ in range(2):
    st.step()
with Spy():
    st.step()
tot = sum(bytes_.values())
print(f"copy-like ops in one step: {sum(log.values())} calls, {tot/1e6:.1f} MB written")
for key, b in bytes_.most_common(70):
    print(f"{b/1e6:9.2f} MB x{log[key]:3d}  {key[0]:10s} {key[1][:110]:110s} {key[2][:45]:45s} {key[3]}")
