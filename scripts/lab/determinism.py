#!/usr/bin/env python3
"""Is one forward/backward of the R1 step reproducible call-to-call (same weights, frame, RNG seed)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401
import torch
from omnihd_amd.harness import FusionTrainStep
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype=dt, miopen_find=False)
for _ in range(4):
    st.step()
torch.cuda.synchronize()
m = st.raw_model
runs = []
for trial in range(4):
    b = st.batches[0]
    st.opt.zero_grad(set_to_none=True)
    torch.manual_seed(123)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == "bf16"):
        losses = st.model(return_loss=True, **b)
    total = sum(v if torch.is_tensor(v) else sum(v) for v in losses.values())
    total.backward()
    torch.cuda.synchronize()
    runs.append(({k: [round(float(x), 6) for x in (v if isinstance(v, (list, tuple)) else [v])] for k, v in losses.items()},
                 {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}))
    print(trial, runs[-1][0], flush=True)
for i in range(1, len(runs)):
    worst = sorted(((float((runs[i][1][n] - g).abs().max() / g.abs().max().clamp_min(1e-20)), n) for n, g in runs[0][1].items()), reverse=True)
    print("run", i, "vs 0: params with deviation > 1e-3:", sum(1 for d, _ in worst if d > 1e-3), "of", len(worst))
    if i == 1:
        for d, n in worst[:40]:
            print(f"   {d:9.2e}  |g|max {float(runs[0][1][n].abs().max()):9.2e}  {n}  {tuple(runs[0][1][n].shape)}")
