"""Host-side helpers of round 5 that need no GPU: the per-step cache of environment switches (omnihd_amd/_env.py) and the
guarded single-GPU launch of bench.py (a child that exits on its own is not restarted; only a signal death is)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]


def test_env_is_live_outside_an_epoch_and_cached_inside(monkeypatch):
    from omnihd_amd import _env
    monkeypatch.delenv("OMNIHD_TEST_SWITCH", raising=False)
    assert _env.env("OMNIHD_TEST_SWITCH", "d") == "d"
    monkeypatch.setenv("OMNIHD_TEST_SWITCH", "1")
    assert _env.env("OMNIHD_TEST_SWITCH", "d") == "1"            # outside an epoch: every call asks os.environ
    _env.epoch_begin()
    try:
        assert _env.env("OMNIHD_TEST_SWITCH", "d") == "1"
        monkeypatch.setenv("OMNIHD_TEST_SWITCH", "2")
        assert _env.env("OMNIHD_TEST_SWITCH", "d") == "1"        # inside: looked up once per step
        monkeypatch.delenv("OMNIHD_TEST_OTHER", raising=False)
        assert _env.env("OMNIHD_TEST_OTHER", "dflt") == "dflt"   # an unset switch is cached as unset, the default still applies
        assert _env.env("OMNIHD_TEST_OTHER", "x") == "x"
    finally:
        _env.epoch_end()
    assert _env.env("OMNIHD_TEST_SWITCH", "d") == "2"            # the next step sees the new value
    _env.epoch_begin()
    _env.epoch_begin()                                           # nested (a step inside a step-like scope): one cache, closed last
    _env.epoch_end()
    monkeypatch.setenv("OMNIHD_TEST_SWITCH", "3")
    assert _env.env("OMNIHD_TEST_SWITCH", "d") in ("2", "3")
    _env.epoch_end()
    assert _env.env("OMNIHD_TEST_SWITCH", "d") == "3"


def test_bench_guard_does_not_restart_a_child_that_exits_by_itself():
    """No GPU here: the child says so and exits 1; the guard reports one attempt (a restart is for signal deaths only)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OMNIHD_BENCH_CHILD")}
    env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert out.stderr.count("bench.py needs a GPU") == 1, out.stderr[-800:]
    assert "attempt 1 ended with exit code 1" in out.stderr and "starting it once more" not in out.stderr
