"""The ablation / trace builds of the lab scripts are PATCHES against the product kernels (scripts/lab/patches/*.patch, ADVICE round 3:
"keep the diffs, not the copies").  A patch that no longer applies has rotted: this test keeps them honest."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


PATCHES = sorted(glob.glob(os.path.join(ROOT, "scripts", "lab", "patches", "*.patch")) +
                 glob.glob(os.path.join(ROOT, "scripts", "lab", "patches", "on_superseded", "*.patch")))


@pytest.mark.parametrize("patch", PATCHES, ids=lambda p: os.path.relpath(p, os.path.join(ROOT, "scripts", "lab", "patches")))
def test_lab_patch_applies_to_the_product_source(patch, tmp_path):
    """``on_superseded/*.patch`` (hooks of kernels that left the product in round 6) apply on top of
    ``pool_superseded_kernels.patch``, which restores those kernels."""
    if shutil.which("patch") is None:
        pytest.skip("no patch(1) here")
    head = open(patch).readline()
    m = re.match(r"--- (\S*csrc/(\w+\.hip))", head)
    assert m, head
    target = os.path.join(ROOT, "omnihd-scenes_amd", "csrc", m.group(2))
    if os.path.basename(os.path.dirname(patch)) == "on_superseded":
        base = os.path.join(ROOT, "scripts", "lab", "patches", "pool_superseded_kernels.patch")
        out = subprocess.run(["patch", "-s", "-o", str(tmp_path / "restored.hip"), target, base], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        target = str(tmp_path / "restored.hip")
    out = subprocess.run(["patch", "-s", "-o", str(tmp_path / "patched.hip"), target, patch], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OMNIHD_" in open(tmp_path / "patched.hip").read()


def test_superseded_kernels_patch_restores_the_removed_kernels(tmp_path):
    if shutil.which("patch") is None:
        pytest.skip("no patch(1) here")
    target = os.path.join(ROOT, "omnihd-scenes_amd", "csrc", "bev_pool_v2.hip")
    src = open(target).read()
    for gone in ("k_pool_fwd_tiles", "k_pool_fwd_lean2", "k_pool_bwd_sched", "k_pool_bwd_stream"):
        assert "void %s(" % gone not in src, gone + " is back in the product source"
    assert len(src.splitlines()) <= 1000
    base = os.path.join(ROOT, "scripts", "lab", "patches", "pool_superseded_kernels.patch")
    out = subprocess.run(["patch", "-s", "-o", str(tmp_path / "restored.hip"), target, base], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    restored = open(tmp_path / "restored.hip").read()
    for back in ("k_pool_fwd_tiles", "k_pool_fwd_lean2", "k_pool_bwd_sched", "k_pool_bwd_stream"):
        assert "void %s(" % back in restored, back
