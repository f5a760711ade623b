#!/bin/bash
# Round 6: fp32 bench line with the per_frame_calibration block
export TMPDIR=/tmp; out=gpurun_out/r6_02; mkdir -p $out
OMNIHD_BENCH_DDP1=0 timeout 1200 python3 bench.py --dtype fp32 --no-cpu-baseline > $out/bench_fp32.json 2> $out/bench_fp32.err; echo "bench rc $?"
tail -3 $out/bench_fp32.err | cut -c1-300
python3 - <<PY
import json
l = json.loads(open("$out/bench_fp32.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value", "ms_per_step", "step_ms")})
print("per_frame", {k: v for k, v in l["per_frame_calibration"].items() if k != "note"})
print("plan_build", {k: v for k, v in l["ops_roofline"]["plan_build"].items() if k != "note"})
print("roofline", {k: l["roofline"][k] for k in ("mean_launch_us", "frac", "bwd_mean_launch_us", "bwd_frac")})
print("fast", l["fast_paths"])
PY
