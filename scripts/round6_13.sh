#!/bin/bash
# Round 6: the whole GPU suite on the final tree + exposure runs of the default step (radar branch in line)
export TMPDIR=/tmp; out=gpurun_out/r6_13; mkdir -p $out
timeout 2400 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -8 $out/pytest_gpu.txt
for i in 1 2 3; do
  timeout 600 python3 scripts/lab/fault_repro.py bf16 500 > $out/expo_$i.out 2> $out/expo_$i.err; echo "exposure bf16 measured choices, default switches, run $i rc $?: $(tail -1 $out/expo_$i.out)"
  rm -f $out/expo_$i.err
done
