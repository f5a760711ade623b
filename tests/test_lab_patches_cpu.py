"""The ablation / trace builds of the lab scripts are PATCHES against the product kernels (scripts/lab/patches/*.patch, ADVICE round 3:
"keep the diffs, not the copies").  A patch that no longer applies has rotted: this test keeps them honest."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("patch", sorted(glob.glob(os.path.join(ROOT, "scripts", "lab", "patches", "*.patch"))), ids=os.path.basename)
def test_lab_patch_applies_to_the_product_source(patch, tmp_path):
    if shutil.which("patch") is None:
        pytest.skip("no patch(1) here")
    head = open(patch).readline()
    m = re.match(r"--- (\S*csrc/(\w+\.hip))", head)
    assert m, head
    target = os.path.join(ROOT, "omnihd-scenes_amd", "csrc", m.group(2))
    out = subprocess.run(["patch", "-s", "-o", str(tmp_path / "patched.hip"), target, patch], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OMNIHD_" in open(tmp_path / "patched.hip").read()
