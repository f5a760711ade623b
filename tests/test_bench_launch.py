"""`python bench.py --gpus N` starts its N ranks itself (the reference launches N ranks from one command:
tools/dist_train.sh:7-9).  CPU: the launch / rendezvous / rank-0-reports skeleton over gloo; GPU box (one device): two ranks
sharing cuda:0 over gloo through the real bev_ops workload."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def _json_line(out):
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks_and_rank0_prints_one_line():
    out = _run(["--gpus", "2", "--selftest-launch", "--steps", "3", "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = _json_line(out)
    assert line["n_gpus"] == 2 and line["max_over_ranks"] == 2.0 and line["steps"] == 3 and line["warmup"] == 1
    assert line["comm"] == {"world_size": 2, "backend": "gloo"}       # the process group's own account of the job (not --gpus echoed)


def test_world_size_must_match_gpus():
    out = _run(["--gpus", "2", "--selftest-launch"], env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr


def test_a_failing_rank_fails_the_command():
    out = _run(["--gpus", "2", "--workload", "bev_ops", "--steps", "1", "--warmup", "0"], env={"OMNIHD_BENCH_SHARE_GPU": "1",
                                                                                             "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert out.returncode != 0          # no GPU: every rank exits with "bench.py needs a GPU", the launcher reports failure


@pytest.mark.gpu
def test_gpus_2_sharing_one_device_reports_two_ranks(cuda):
    out = _run(["--gpus", "2", "--workload", "bev_ops", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                "--kernel-launches", "10"], env={"OMNIHD_BENCH_SHARE_GPU": "1"}, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0
    assert line["scaling"] == "weak" and "roofline" in line and "ops_roofline" in line
