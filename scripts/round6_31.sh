#!/bin/bash
# Round 6 (last GPU seconds): A/B of the fp32 step with the hi / lo hand-over also from the frozen-BatchNorm epilogue
export TMPDIR=/tmp; out=gpurun_out/r6_31; mkdir -p $out
for rep in 1 2; do for ho in 1 all; do
OMNIHD_SPLIT_HANDOVER=$ho timeout 40 python3 scripts/lab/fault_repro.py fp32 40 > $out/step_$ho.txt 2> $out/step_$ho.err; echo "handover=$ho rc $? $(tail -1 $out/step_$ho.txt)"
done; done
