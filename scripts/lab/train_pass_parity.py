#!/usr/bin/env python3
"""Report (no assertions) of tests/test_detector_gpu.py::test_full_size_r1_training_pass_gradients...: relative L2 of outputs and
gradients, GPU (split / miopen policy) vs the oracle-backed CPU run, and GPU vs GPU (split vs miopen, and the same policy twice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from tests.test_detector_gpu import _train_pass, GRAD_NAMES
torch.backends.cudnn.allow_tf32 = False
OPEN = len(sys.argv) > 1 and sys.argv[1] == "open"
print("relu_open =", OPEN)
runs = {}
for pol in ("split", "miopen", "split"):
    os.environ["OMNIHD_FP32_CONV"] = pol
    runs.setdefault(pol, []).append(_train_pass("cuda:0", False, OPEN))
cpu = _train_pass("cpu", True, OPEN)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
def report(name, a, b):
    keys = ["depth", "bev", "reg"]
    print(name, " ".join(f"{k} {rel(a[k], b[k]):.1e}" for k in keys), "| losses", {k: f"{a['losses'][k]:.6f}/{b['losses'][k]:.6f}" for k in b["losses"]},
          f"depth_loss {a['depth_loss']:.6f}/{b['depth_loss']:.6f}")
    for n in GRAD_NAMES:
        print(f"    grad {n:70s} {rel(a['grads'][n], b['grads'][n]):.2e}   |g| {float(b['grads'][n].norm()):.3e}")
report("split vs cpu ", runs["split"][0], cpu)
report("miopen vs cpu", runs["miopen"][0], cpu)
report("split vs miopen (gpu)", runs["split"][0], runs["miopen"][0])
report("split vs split (gpu, run to run)", runs["split"][1], runs["split"][0])
