#!/bin/bash
# Round-5 GPU call 35: final validation — whole GPU suite, smoke(), default bench line.
export TMPDIR=/tmp; out=gpurun_out/r5final; mkdir -p $out
timeout 3000 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 > $out/pytest_gpu.txt; cat $out/pytest_gpu.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $out/smoke.txt; cat $out/smoke.txt
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
d=json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print("fp32", d["value"], d["ms_per_step"], d["step_ms"]); print("bf16", d["bf16_autocast"]["value"], d["bf16_autocast"]["ms_per_step"], d["bf16_autocast"]["step_ms"]); print("ddp", d["ddp_1rank"]["ms_per_step"], d["ddp_1rank"]["overhead_vs_plain"]); print("roofline", d["roofline"]["frac"], d["roofline"].get("frac_vs_copy_peak"), d["roofline"]["mean_launch_us"])
PY
