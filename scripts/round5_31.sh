#!/bin/bash
# Round-5 GPU call 31: cached environment switches + detector test without the early return: tests, bench.
export TMPDIR=/tmp; out=gpurun_out/r5ae; mkdir -p $out
timeout 1800 python3 -m pytest tests/test_detector_gpu.py tests/test_determinism_gpu.py tests/test_bench_launch.py -m gpu -q -x 2>&1 | tail -6 > $out/tests.txt; cat $out/tests.txt
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
d=json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print("fp32", d["ms_per_step"], d["step_ms"]); print("bf16", d["bf16_autocast"]["ms_per_step"], d["bf16_autocast"]["step_ms"]); print("ddp", d["ddp_1rank"]["ms_per_step"], d["ddp_1rank"]["overhead_vs_plain"])
PY
