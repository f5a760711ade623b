#!/bin/bash
# Round-5 GPU call 26: where is the GPU idle?  plain fp32 step vs the same step in a one-rank DDP group.
export TMPDIR=/tmp; out=gpurun_out/r5z; mkdir -p $out
STEP_PROFILE_GAPS=1 STEP_PROFILE_OUT=$out/plain bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_gaps.txt 2>&1; grep -A28 "GPU idle" $out/step_fp32_gaps.txt | cut -c1-170
STEP_PROFILE_GAPS=1 OMNIHD_STEP_DDP=1 STEP_PROFILE_WARM=5 STEP_PROFILE_OUT=$out/ddp bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_ddp1_gaps.txt 2>&1; grep -A28 "GPU idle" $out/step_fp32_ddp1_gaps.txt | cut -c1-170
