#!/usr/bin/env python3
"""Tiny detector, fp32: gradients on the GPU (HIP operators, split / miopen convolution policy) vs the CPU run over the oracle
operators — per tensor: max-abs error over the largest entry, and relative L2 error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd"), os.path.join(ROOT, "tests")]
import torch
import test_detector_gpu as T
cpu = T._run("cpu", use_oracle=True)
for policy in ("split", "miopen"):
    os.environ["OMNIHD_FP32_CONV"] = policy
    gpu = T._run("cuda:0", use_oracle=False)
    rows = []
    for n, b in cpu["grads"].items():
        a = gpu["grads"][n].double(); b = b.double()
        mx = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
        l2 = float((a - b).norm()) / max(float(b.norm()), 1e-12)
        rows.append((mx, l2, n))
    rows.sort(reverse=True)
    print(f"policy {policy}: {len(rows)} gradient tensors; max-abs/max: worst {rows[0][0]:.2e}, over 1e-3: {sum(r[0] > 1e-3 for r in rows)}; "
          f"relative L2: worst {max(r[1] for r in rows):.2e}, over 1e-3: {sum(r[1] > 1e-3 for r in rows)}")
    for mx, l2, n in rows[:12]:
        print(f"   {mx:.2e}  L2 {l2:.2e}  {n}")
