#!/usr/bin/env python3
"""cProfile of the host side of the R1 training step: python3 scripts/lab/step_cprofile.py [bf16|fp32] [steps]"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401
import torch
from omnihd_amd.harness import FusionTrainStep
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
os.environ["OMNIHD_DUAL_STREAM"] = os.environ.get("OMNIHD_DUAL_STREAM", "1")
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype=dt, miopen_find=True)
for _ in range(8):
    st.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    st.step()
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr)
ps.sort_stats("tottime")
print("per step (ms), %d steps" % n)
rows = sorted(ps.stats.items(), key=lambda kv: -kv[1][2])[:45]
for (f, line, name), (cc, nc, tt, ct, _) in rows:
    print("%7.3f tot %7.3f cum %6.0f calls  %s:%d %s" % (tt / n * 1e3, ct / n * 1e3, nc / n, os.path.basename(f), line, name))
