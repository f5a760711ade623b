"""configs[3] on a one-GPU box: two ranks SHARING cuda:0 (gloo; RCCL refuses two ranks on one device) run the tiny detector's
training step through the HIP kernels — DDP's gradient all-reduce, the naiveSyncBN statistic exchanges of the fused BatchNorm
kernels and of the fused pillar feature net (mean of rank means, the reference's ops/norm.py:55-82), the dual-stream forward —
and must end with bit-identical replicas although each rank saw different frames.  The same script runs the full R1 model
(scripts/lab/ddp_shared_gpu.py r1)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_keep_replicas_identical_through_the_hip_path(cuda):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "lab", "ddp_shared_gpu.py"), "tiny", "2"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    last = [ln for ln in out.stdout.splitlines() if ln.startswith("rank ") and "replicas identical" in ln]
    assert len(last) == 2 and all("replicas identical True" in ln for ln in last), out.stdout[-2000:]
    assert "'world_size': 2" in last[0] and "'syncbn_exchanges_per_step'" in last[0]
    assert out.stdout.strip().endswith("OK")
