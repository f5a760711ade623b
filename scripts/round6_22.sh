#!/bin/bash
# Round 6: is the TF32-grade step bound by the host?  (busy-wait added to every library call)
export TMPDIR=/tmp; out=gpurun_out/r6_22; mkdir -p $out
OMNIHD_FP32_CONV=f16 timeout 300 python3 scripts/lab/host_slack.py fp32 > $out/host_slack_f16.txt 2> $out/host_slack_f16.err; echo "rc $?"; cat $out/host_slack_f16.txt
timeout 200 python3 -m pytest tests/test_conv_f16_gpu.py -x -q -p no:cacheprovider -k "saturates or temporary" 2>&1 | tail -2
