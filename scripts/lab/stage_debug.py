import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from tests import test_stage_gradients_gpu as T
os.environ["OMNIHD_FP32_CONV"] = sys.argv[1] if len(sys.argv) > 1 else "miopen"
gpu, cpu = T._models()
T._open_relus(gpu, True); T._open_relus(cpu, True)
make, shape = T._stages(gpu)["depthnet_heads"]
got = T._run(gpu, make, shape, "cuda:0"); want = T._run(cpu, make, shape, "cpu")
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
for n in want["grads"]:
    r = rel(got["grads"][n], want["grads"][n])
    if r > 1e-3:
        print(f"{r:.2e} |got| {float(got['grads'][n].norm()):.3e} |want| {float(want['grads'][n].norm()):.3e}  {n}")
