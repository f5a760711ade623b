#!/bin/bash
# Round 6: first GPU run of the TF32-grade half form — its parity tests, then the R1 fp32 step under the default policy and under f16
export TMPDIR=/tmp; out=gpurun_out/r6_17; mkdir -p $out
timeout 600 python3 -m pytest tests/test_conv_f16_gpu.py -x -q -s -p no:cacheprovider > $out/pytest_f16.txt 2>&1; echo "pytest rc $?"
tail -25 $out/pytest_f16.txt
for pol in tune f16; do
  OMNIHD_FP32_CONV=$pol timeout 300 python3 scripts/lab/fault_repro.py fp32 40 > $out/step_$pol.txt 2> $out/step_$pol.err; echo "$pol rc $?"; tail -1 $out/step_$pol.txt
done
OMNIHD_FP32_CONV=f16 STEP_PROFILE_OUT=$out/f16 bash scripts/lab/step_profile.sh fp32 6 > $out/step_f16_steady.txt 2>&1
find $out -name "*.csv" -size +1M -delete
head -40 $out/step_f16_steady.txt
