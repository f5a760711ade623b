#!/bin/bash
# Round-5 GPU call 24: new choice table + per-tap weight images: tests that cover them, fp32 bench, step outliers.
export TMPDIR=/tmp; out=gpurun_out/r5x; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_conv_split_gpu.py tests/test_bench_launch.py tests/test_determinism_gpu.py -m gpu -q -x 2>&1 | tail -6 > $out/tests.txt; cat $out/tests.txt
python3 bench.py --dtype fp32 > $out/bench_fp32.json 2> $out/bench_fp32.err; cut -c1-300 $out/bench_fp32.json
timeout 600 python3 scripts/lab/step_outliers.py fp32 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/step_outliers.txt; grep -v "^   step" $out/step_outliers.txt; grep "^   step" $out/step_outliers.txt | sort -k3 -n -r | head -8
STEP_PROFILE_OUT=$out/fp32 bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_steady.txt 2>&1; head -12 $out/step_fp32_steady.txt | cut -c1-150
