#!/bin/bash
# Round-5 last GPU call: bench-launch tests with the final bench.py, then the final default line.
export TMPDIR=/tmp; out=gpurun_out/r5last; mkdir -p $out
timeout 1800 python3 -m pytest tests/test_bench_launch.py tests/test_wgrad_nhwc_gpu.py -m gpu -q 2>&1 | tail -3 > $out/tests.txt; cat $out/tests.txt
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc $?"
grep "bench.py" $out/bench_default.err | tail -14
python3 - <<PY
import json
d=json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print("fp32", d["value"], d["ms_per_step"], d["step_ms"]); print("bf16", d["bf16_autocast"]["value"], d["bf16_autocast"]["ms_per_step"], d["bf16_autocast"]["step_ms"]); x=d["ddp_1rank"]; print("ddp", {k:x[k] for k in x if k.startswith(("ms_","plain_after_ms","overhead"))}); print("fresh", json.dumps(x.get("fresh_process"))[:500]); print("roofline", d["roofline"]["frac"], d["roofline"].get("frac_vs_copy_peak"), d["roofline"]["mean_launch_us"]); print("cpu", d["cpu_baseline"]["value"])
PY
