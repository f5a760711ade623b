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


def test_bench_runs_its_measurement_in_one_child_and_never_retries():
    """No GPU here: the measurement child says so and exits 1; the parent passes that exit code on.  Round 6: nothing is retried
    any more, also not a child that dies of a signal (the GPU memory fault rounds 4-5 retried around was found and removed)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OMNIHD_BENCH_CHILD")}
    env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert out.stderr.count("bench.py needs a GPU") == 1, out.stderr[-800:]
    assert "the measurement process ended with exit code 1" in out.stderr and "once more" not in out.stderr
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "OMNIHD_BENCH_SAFE" not in src and "attempt" not in src.split("def run_guarded")[1].split("def launch_ranks")[0]


def test_nhwc_weight_gradient_plan_fills_the_chip_and_is_not_clamped_by_the_slab_budget():
    """Host side of csrc/conv_wgrad_nhwc.hip (no GPU work): the split-K plan behind the workspace size.  Round 5's first version
    capped the slabs at 64 MB and silently ran 1024 -> 1024 3x3 at 160 x 240 with ONE split (192 workgroups on 256 CUs) and
    640 -> 384 with 7 ragged ones; the cap is 256 MB now."""
    from omnihd_amd._lib import lib
    L = lib()
    ws = L.omnihd_conv_wgrad_nhwc_workspace_bytes

    def splits(B, H, W, cin, cout, k, s, p, d):
        Ho, Wo = (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1
        n = ws(B, H, W, cin, Ho, Wo, cout, k, s, p, d)
        assert n >= 256
        return round((n - 256) / (cout * k * k * cin * 4))

    # three-taps form, one workgroup per CU: (Cout tiles x Cin tiles x 3 kernel rows) x splits is a whole number of rounds of 256
    assert splits(1, 160, 240, 1024, 1024, 3, 1, 1, 1) == 4          # 192 groups x 4 = 768 = 3 rounds
    assert splits(1, 160, 240, 640, 384, 3, 1, 1, 1) == 11           # 45 groups x 11 = 495
    assert splits(6, 64, 176, 256, 256, 3, 1, 1, 1) == 21            # 12 groups x 21 = 252
    # one split: no slab at all
    assert ws(1, 2, 3, 8, 2, 3, 8, 3, 1, 1, 1) == 256
    # not taken: 5x5, channel counts that are not multiples of 8, inconsistent output size
    assert ws(1, 64, 64, 64, 64, 64, 64, 5, 1, 2, 1) == 0
    assert ws(1, 64, 64, 60, 64, 64, 64, 3, 1, 1, 1) == 0
    assert ws(1, 64, 64, 64, 63, 64, 64, 3, 1, 1, 1) == 0
