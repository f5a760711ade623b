#!/usr/bin/env python3
"""Inference frames/s with the protocol of the reference's own tool (tools/analysis_tools/benchmark.py:70-98): eval
mode, one frame per call through ``model(return_loss=False, rescale=True, ...)``, a device synchronisation before and
after EVERY call, the first 5 calls skipped, mean over the rest.  Synthetic frames (the dataset is unreachable), random
weights, bf16 autocast for the dense layers unless ``fp32`` is given.
    python scripts/infer_fps.py [r1|r2] [bf16|fp32] [samples=200] [det|camera]
``camera`` = BASELINE.json configs[1], the camera-only detector of projects/configs/bevfusion_NewScenes/cam_stream/LSS.py.
Not part of the bench contract; results: profiles/round3/infer_fps.txt."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch  # noqa: E402

from omnihd_amd.harness import FusionTrainStep  # noqa: E402

res = sys.argv[1] if len(sys.argv) > 1 else "r1"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
samples = int(sys.argv[3]) if len(sys.argv) > 3 else 200
task = sys.argv[4] if len(sys.argv) > 4 else "det"
st = FusionTrainStep(res=res, batch=1, radar_dims=7 if res == "r1" else 8, dtype=dtype, sets=4, miopen_find=True, task=task)
model = st.raw_model.eval()
torch.nn.init.constant_(model.pts_bbox_head.conv_cls.bias, -2.0)          # random weights: keep the NMS input realistic
num_warmup, pure = 5, 0.0
for i in range(samples):
    b = st.batches[i % len(st.batches)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
        out = model(return_loss=False, rescale=True, points=[b["points"]], img_metas=[b["img_metas"]], img=[b["img"]])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if i >= num_warmup:
        pure += dt
        if (i + 1) % 50 == 0:
            print(f"Done frame [{i + 1:<3}/ {samples}], fps: {(i + 1 - num_warmup) / pure:.1f} frames / s", flush=True)
print(f"Overall fps: {(samples - num_warmup) / pure:.1f} frames / s  ({task}, {res}, {dtype}, {len(out[0]['pts_bbox']['boxes_3d'])} boxes in the last frame)")
