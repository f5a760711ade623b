#!/bin/bash
# Round-5 GPU call 21: whole GPU suite, default bench line, fp32 bench line, steady-state profiles.
export TMPDIR=/tmp; out=gpurun_out/r5u; mkdir -p $out
timeout 2400 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $out/pytest_gpu.txt; cat $out/pytest_gpu.txt
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; cut -c1-600 $out/bench_default.json
python3 bench.py --dtype fp32 > $out/bench_fp32.json 2> $out/bench_fp32.err; cut -c1-300 $out/bench_fp32.json
STEP_PROFILE_OUT=$out/fp32 bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_steady.txt 2>&1; head -30 $out/step_fp32_steady.txt
STEP_PROFILE_OUT=$out/bf16 bash scripts/lab/step_profile.sh bf16 6 > $out/step_bf16_steady.txt 2>&1; head -24 $out/step_bf16_steady.txt
