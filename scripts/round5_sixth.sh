#!/bin/bash
# Round-5 sixth GPU call: anchor loss + voxeliser tests, voxeliser timing, deterministic step (no find mode), copies census.
export TMPDIR=/tmp; out=gpurun_out/r5f; mkdir -p $out
timeout 900 python3 -m pytest tests/test_anchor_loss_gpu.py tests/test_radar_gpu.py tests/test_radar_properties_gpu.py tests/test_determinism_gpu.py -m gpu -q 2>&1 | tail -30 > $out/tests.txt; cat $out/tests.txt
timeout 300 python3 - <<'PY' 2>&1 | grep -v "^/opt" | tee $out/voxelize_time.txt
import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "omnihd-scenes_amd")]
import numpy as np, torch
from omnihd_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
n = 19753
p = np.concatenate([rng.uniform(-60, 60, (n, 1)), rng.uniform(-40, 40, (n, 1)), rng.uniform(-3, 5, (n, 1)), rng.standard_normal((n, 4))], 1).astype(np.float32)
pts = torch.from_numpy(p).to(dev)
for grid in ("1", "0"):
    os.environ["OMNIHD_VOXELIZE_GRID"] = grid
    fn = lambda: ops.hard_voxelize_async(pts, [0.25, 0.25, 8], [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0], 10, 30000)
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(8_000_000)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print("OMNIHD_VOXELIZE_GRID=%s: %.1f us of device time per call (back to back)" % (grid, e0.elapsed_time(e1) * 1e3 / 50))
PY
OMNIHD_DETERMINISTIC=1 timeout 600 python3 bench.py --dtype fp32 --no-cpu-baseline > $out/bench_deterministic.json 2> $out/bench_deterministic.err; cut -c1-330 $out/bench_deterministic.json
timeout 600 python3 scripts/trace_copies.py fp32 2>&1 | grep -v "^/opt\|Warn\|warn" | head -45 > $out/copies_fp32.txt; head -30 $out/copies_fp32.txt
