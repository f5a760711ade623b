#!/bin/bash
# Round 6: what the host spends a step on (TF32-grade fp32 step and the bf16 step)
export TMPDIR=/tmp; out=gpurun_out/r6_19; mkdir -p $out
OMNIHD_FP32_CONV=f16 timeout 300 python3 scripts/lab/host_profile.py fp32 10 > $out/host_f16.txt 2> $out/host_f16.err; echo "rc $?"; head -30 $out/host_f16.txt
