#!/bin/bash
# Round 6: the half form after the amax fix (one atomic per workgroup) and the ring of scale slots; where the strided copies come from
export TMPDIR=/tmp; out=gpurun_out/r6_18; mkdir -p $out
timeout 600 python3 -m pytest tests/test_conv_f16_gpu.py -x -q -p no:cacheprovider > $out/pytest_f16.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_f16.txt
OMNIHD_FP32_CONV=f16 timeout 300 python3 scripts/lab/fault_repro.py fp32 40 > $out/step_f16.txt 2> $out/step_f16.err; echo "f16 rc $?"; tail -1 $out/step_f16.txt
OMNIHD_FP32_CONV=f16 STEP_PROFILE_OUT=$out/f16 bash scripts/lab/step_profile.sh fp32 6 > $out/step_f16_steady.txt 2>&1
find $out -name "*.csv" -size +1M -delete
head -24 $out/step_f16_steady.txt
OMNIHD_FP32_CONV=f16 timeout 300 python3 scripts/lab/copy_hunt.py fp32 3 > $out/copy_hunt_f16.txt 2> $out/copy_hunt_f16.err; echo "hunt rc $?"; head -50 $out/copy_hunt_f16.txt
