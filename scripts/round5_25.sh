#!/bin/bash
# Round-5 GPU call 25: the fp32 step inside a one-rank process group under DistributedDataParallel: kernel statistics beside the plain step's.
export TMPDIR=/tmp; out=gpurun_out/r5y; mkdir -p $out
OMNIHD_STEP_DDP=1 STEP_PROFILE_WARM=5 STEP_PROFILE_OUT=$out/ddp bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_ddp1.txt 2>&1; head -60 $out/step_fp32_ddp1.txt | cut -c1-150
