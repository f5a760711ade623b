#!/bin/bash
# Round-5 GPU call 15: counters of the three-taps weight gradient; steady-state fp32 + bf16 step profiles with the NHWC kernel everywhere.
export TMPDIR=/tmp; out=gpurun_out/r5o; mkdir -p $out
bash scripts/lab/pmc_wgrad.sh $out/pmc_1024 1,160,240,1024,1024,3,1,1 > $out/pmc_wgrad_1024.txt 2>&1; cat $out/pmc_wgrad_1024.txt
bash scripts/lab/pmc_wgrad.sh $out/pmc_256 6,64,176,256,256,3,1,1 > $out/pmc_wgrad_256.txt 2>&1; cat $out/pmc_wgrad_256.txt
STEP_PROFILE_OUT=$out/fp32 bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_steady.txt 2>&1; head -50 $out/step_fp32_steady.txt
STEP_PROFILE_OUT=$out/bf16 bash scripts/lab/step_profile.sh bf16 6 > $out/step_bf16_steady.txt 2>&1; head -40 $out/step_bf16_steady.txt
python3 bench.py --dtype fp32 > $out/bench_fp32.json 2> $out/bench_fp32.err; cut -c1-400 $out/bench_fp32.json
