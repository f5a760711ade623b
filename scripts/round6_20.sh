#!/bin/bash
# Round 6: the gates of the TF32-grade form (kernel parity, TF32-emulation comparison at R1 + R2, stage gradients), its step time after the host trims
export TMPDIR=/tmp; out=gpurun_out/r6_20; mkdir -p $out
timeout 900 python3 -m pytest tests/test_conv_f16_gpu.py -x -q -s -p no:cacheprovider > $out/pytest_f16.txt 2>&1; echo "pytest rc $?"; grep -a "relative L2\|max |diff\|loss traj\|passed\|failed" $out/pytest_f16.txt
timeout 900 python3 -m pytest tests/test_stage_gradients_gpu.py -q -s -p no:cacheprovider -k "f16" > $out/pytest_stage_f16.txt 2>&1; echo "stage rc $?"; grep -a "STAGE\|passed\|failed\|Error" $out/pytest_stage_f16.txt
OMNIHD_FP32_CONV=f16 timeout 300 python3 scripts/lab/fault_repro.py fp32 40 > $out/step_f16.txt 2> $out/step_f16.err; echo "f16 rc $?"; tail -1 $out/step_f16.txt
OMNIHD_FP32_CONV=f16 timeout 300 python3 scripts/lab/host_profile.py fp32 10 > $out/host_f16.txt 2> $out/host_f16.err; echo "rc $?"; head -2 $out/host_f16.txt
