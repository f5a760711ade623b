#!/bin/bash
# Round 6: the half plane from the producers' epilogues and the amax from their backward — tests, A/B of the step on one box, profile
export TMPDIR=/tmp; out=gpurun_out/r6_23; mkdir -p $out
timeout 900 python3 -m pytest tests/test_conv_f16_gpu.py -x -q -p no:cacheprovider > $out/pytest_f16.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest_f16.txt
timeout 600 python3 -m pytest tests/test_conv_split_gpu.py tests/test_bn_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
for ho in 1 0; do
OMNIHD_F16_HANDOVER=$ho OMNIHD_FP32_CONV=f16 timeout 300 python3 scripts/lab/fault_repro.py fp32 40 > $out/step_f16_$ho.txt 2> $out/step_f16_$ho.err; echo "f16 handover=$ho rc $?"; tail -1 $out/step_f16_$ho.txt
done; done
timeout 300 python3 scripts/lab/fault_repro.py fp32 40 > $out/step_tune.txt 2> $out/step_tune.err; echo "tune rc $?"; tail -1 $out/step_tune.txt
OMNIHD_FP32_CONV=f16 STEP_PROFILE_OUT=$out/f16 bash scripts/lab/step_profile.sh fp32 6 > $out/step_f16_steady.txt 2>&1
find $out -name "*.csv" -size +1M -delete
head -24 $out/step_f16_steady.txt
