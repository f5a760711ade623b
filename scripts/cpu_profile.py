#!/usr/bin/env python3
"""cProfile of the host side of a few steady-state training steps (where does the enqueue time go?)."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd.harness import FusionTrainStep
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype=(sys.argv[1] if len(sys.argv) > 1 else "bf16"), miopen_find=True)
for _ in range(5):
    st.step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)          # backward on this thread so that it is profiled
st.step(); torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    st.step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(38)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[:60]))
