"""A training step that is run-to-run identical, bit for bit (atomics-free, reproducible results; VERDICT round 4
#2).  Round 4 measured 2-3e-2 between two identical runs in the image backbone's weight gradients: the library's fp32 kernels for
strided convolutions and small weight gradients accumulate with atomics.  With OMNIHD_DETERMINISTIC=1 every convolution pass this
library has a kernel for runs on it (csrc/conv_igemm.hip, conv_gen.hip, conv_wgrad.hip: fixed-order sums, no atomics) and the
leftovers are kept off the library's atomic solvers.  Reference step: projects/configs/bevfusion_NewScenes/bevfusion.py (fp32,
AdamW, clip 35)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import hashlib, os, sys, torch
sys.path[:0] = [%r, %r]
from omnihd_amd.harness import FusionTrainStep
torch.backends.cudnn.allow_tf32 = False
res, steps = sys.argv[1], int(sys.argv[2])
st = FusionTrainStep(res=res, batch=1, radar_dims=7, device="cuda:0", seed=21, dtype="fp32", sets=2)
import json
h = hashlib.sha256()
losses, per = [], {}
for it in range(steps):
    losses.append(float(st.step().detach()))
    torch.cuda.synchronize()
    for n, p in st.raw_model.named_parameters():
        if p.grad is not None:
            raw = p.grad.detach().cpu().numpy().tobytes()
            h.update(n.encode()); h.update(raw)
            per["step%%d grad %%s" %% (it, n)] = hashlib.sha1(raw).hexdigest()[:10]
for n, p in st.raw_model.named_parameters():
    h.update(p.detach().cpu().numpy().tobytes())
for n, b in st.raw_model.named_buffers():
    raw = b.detach().cpu().numpy().tobytes()
    h.update(raw)
    per["buffer " + n] = hashlib.sha1(raw).hexdigest()[:10]
print("DIGEST", h.hexdigest(), " ".join(repr(v) for v in losses))
print("PER", json.dumps(per))
''' % (ROOT, os.path.join(ROOT, "omnihd-scenes_amd"))


def _run(res, steps, env_extra):
    env = dict(os.environ, OMNIHD_DETERMINISTIC="1", **env_extra)
    out = subprocess.run([sys.executable, "-c", CODE, res, str(steps)], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("DIGEST")][-1].split()
    per = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("PER ")][-1][4:])
    return (line[1], line[2:]), per


def _same(a, b):
    (da, pa), (db, pb) = a, b
    diff = [k for k in pa if pa[k] != pb.get(k)]
    if diff:
        print("differing tensors (%d of %d): %s" % (len(diff), len(pa), " | ".join(diff)))
    assert da == db, ("losses %s vs %s; first differing tensors: %s" % (da[1], db[1], diff[:12]), len(diff))


def test_two_fp32_r1_training_steps_from_the_same_seed_are_bit_identical(cuda):
    """Two fresh processes, the same seed, two full-size R1 fp32 training steps each (forward + backward + clip + AdamW): every
    gradient of both steps, every parameter and every buffer afterwards hash to the same digest; the losses agree to the last
    digit.  The second process additionally runs the forward's radar branch in line (OMNIHD_DUAL_STREAM=0) and the weight
    gradients in line (OMNIHD_WGRAD_OVERLAP=0): stream placement must not change a bit either."""
    a = _run("r1", 2, {})
    b = _run("r1", 2, {})
    _same(a, b)
    c = _run("r1", 2, {"OMNIHD_DUAL_STREAM": "0", "OMNIHD_WGRAD_OVERLAP": "0"})
    _same(a, c)
