#!/bin/bash
# Round-5 third GPU call: A/B of the row-shift kernel's fill schedule, steady-state step profiles (plain vs one-rank DDP), GPU suite.
export TMPDIR=/tmp; out=gpurun_out/r5c; mkdir -p $out
timeout 600 python3 scripts/lab/conv_rs_ab.py 300,301 2>&1 | grep -v "^/opt" > $out/conv_rs_ab.txt; cat $out/conv_rs_ab.txt
STEP_PROFILE_OUT=$out/plain bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_plain.txt 2>&1; head -24 $out/step_fp32_plain.txt
OMNIHD_STEP_DDP=1 STEP_PROFILE_WARM=5 STEP_PROFILE_OUT=$out/ddp bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_ddp.txt 2>&1; head -30 $out/step_fp32_ddp.txt
tail -3 $out/ddp/run.log
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $out/gputests.txt; cat $out/gputests.txt
