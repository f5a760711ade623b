#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5final2; mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc $?"; tail -3 $out/bench_default.err | cut -c1-300
python3 - <<PY
import json
try:
    d=json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
    print("fp32", d["value"], d["ms_per_step"], d["step_ms"]); print("bf16", d["bf16_autocast"]["value"], d["bf16_autocast"]["ms_per_step"], d["bf16_autocast"]["step_ms"]); print("ddp", d["ddp_1rank"]["ms_per_step"], d["ddp_1rank"]["overhead_vs_plain"]); print("roofline", d["roofline"]["frac"], d["roofline"].get("frac_vs_copy_peak"), d["roofline"]["mean_launch_us"])
except Exception as e:
    print("no line:", e)
PY
