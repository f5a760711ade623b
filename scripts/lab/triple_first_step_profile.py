#!/usr/bin/env python3
"""Where do the minutes of the FIRST step of the bs=2 four-frame configuration go?  cProfile of that step (host side)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch  # noqa: E402

from omnihd_amd.harness import FusionTrainStep, seed_miopen_db  # noqa: E402

seed_miopen_db()
t0 = time.perf_counter()
st = FusionTrainStep(res="r1", batch=2, radar_dims=7, device="cuda:0", dtype="bf16", sets=1, task="triple", frames=4)
print("build %.1f s" % (time.perf_counter() - t0), flush=True)
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
st.step()
torch.cuda.synchronize()
pr.disable()
print("first step %.1f s" % (time.perf_counter() - t0), flush=True)
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
t0 = time.perf_counter()
st.step()
torch.cuda.synchronize()
print("second step %.1f s" % (time.perf_counter() - t0), flush=True)
