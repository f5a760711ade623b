#!/bin/bash
# Round 6: who issues the dense copies / adds / cats of the fp32 step (dispatch-mode spy on both threads)
export TMPDIR=/tmp; out=gpurun_out/r6_21; mkdir -p $out
timeout 300 python3 scripts/lab/copy_hunt2.py fp32 > $out/copies_fp32.txt 2> $out/copies_fp32.err; echo "rc $?"; head -70 $out/copies_fp32.txt
