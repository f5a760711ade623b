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


@pytest.mark.gpu
def test_fp32_bench_line_carries_the_one_rank_ddp_block_and_the_live_fast_paths(cuda):
    """the default line says which fast paths were live (counters, not switches) and carries the same
    step inside a one-rank RCCL group with the naiveSyncBN exchanges timed there.  Short run (2 timed steps): the numbers are
    not measurements, the blocks and their consistency are what is asserted."""
    out = _run(["--dtype", "fp32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--kernel-launches", "5"], timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out)
    fp = line["fast_paths"]
    assert fp["kept_output"]["active"] and fp["direct_fwd"]["active"], fp
    assert not fp["dual_stream"]["active"], "the radar branch runs in line since round 6 (profiles/round6/fault_root_cause.txt)"
    assert fp["wgrad_overlap"]["active"] and fp["wgrad_overlap"]["private_hooks_ok"] and fp["choice_table_misses"] == 0, fp
    d = line["ddp_1rank"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["ms_per_step"] > 0 and d["plain_ms_per_step"] == line["ms_per_step"]
    ov = d["wgrad_overlap"]
    assert ov["active"] and ov["ddp"]["hooked"] and ov["ddp"]["settled"] and ov["ddp"]["direct_writes"] > 0, ov
    sb = d["syncbn_exchange_us"]
    assert sb["exchanges_per_step"] >= 40 and sb["total_us_per_step"] > 0, sb
    r = line["roofline"]
    assert r["copy_peak_measured"] > 3000 and abs(r["frac_vs_copy_peak"] - r["achieved"] / r["copy_peak_measured"]) < 1e-3
    assert "bwd_frac_of_copy_peak" in r and "frac_on_moved_bytes" in r
