#!/usr/bin/env python3
"""Capture the persisted kernel-choice table (omnihd-scenes_amd/kernel_choices/gfx950.json) on an MI355X: run the set-up steps
of every workload bench.py / the GPU tests time (fusion detector at R1 and R2 in fp32 and under bf16 autocast, the camera-only
detector, inference), let the per-geometry measurements run with OMNIHD_TUNE_REPEATS=3, and write the merged table.
Usage (GPU box): OMNIHD_CHOICE_TABLE=off python3 scripts/capture_choice_table.py gpurun_out/gfx950.json"""
import os
import sys

os.environ.setdefault("OMNIHD_TUNE_REPEATS", "3")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
from omnihd_amd.harness import FusionTrainStep, seed_miopen_db

seed_miopen_db()
import torch

from omnihd_amd import ops

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "omnihd-scenes_amd", "kernel_choices", "gfx950.json")
for res, dims in (("r1", 7), ("r2", 8)):
    for dt in ("fp32", "bf16"):
        st = FusionTrainStep(res=res, batch=1, radar_dims=dims, device="cuda:0", seed=1234, dtype=dt, miopen_find=True)
        for _ in range(3):
            st.step()
        m = st.raw_model
        m.eval()
        b = st.batches[0]
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == "bf16"):
            m.extract_feat(b["points"], img=b["img"], img_metas=b["img_metas"])
        torch.cuda.synchronize()
        print(res, dt, "choices so far:", len(ops.conv_choices()), len(ops.wgrad_choices()), len(ops.split_choices()), flush=True)
        del st, m
        torch.cuda.empty_cache()
n = ops.save_choice_table(out, note="kernel choices per convolution geometry measured on MI355X (gfx950) by scripts/capture_choice_table.py "
                          "(OMNIHD_TUNE_REPEATS=3): fusion detector at R1 / R2, fp32 (split vs MIOpen) and bf16 (hip vs MIOpen), training + inference")
print("wrote", n, "entries to", out, "| misses (= measured here):", ops.choice_table_info()["misses"])
