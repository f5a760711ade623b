#!/bin/bash
# Round-5 final default bench line with the final bench.py
export TMPDIR=/tmp; out=gpurun_out/r5final4; mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc $?"
grep "bench.py" $out/bench_default.err | tail -12
python3 - <<PY
import json
d=json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print("fp32", d["value"], d["ms_per_step"], d["step_ms"]); print("bf16", d["bf16_autocast"]["value"], d["bf16_autocast"]["ms_per_step"], d["bf16_autocast"]["step_ms"]); print("ddp", d["ddp_1rank"]["ms_per_step"], d["ddp_1rank"]["overhead_vs_plain"]); print("roofline", d["roofline"]["frac"], d["roofline"].get("frac_vs_copy_peak"), d["roofline"]["mean_launch_us"]); print("cpu", d["cpu_baseline"]["value"])
PY
