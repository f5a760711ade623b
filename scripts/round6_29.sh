#!/bin/bash
# Round 6: the fp32-grade step with the producers' hi / lo planes hand-over (OMNIHD_SPLIT_HANDOVER), A/B on one box — measured slower in
# round 4 when a second stream ran beside it; the step is launch-bound in places, so once more
export TMPDIR=/tmp; out=gpurun_out/r6_29; mkdir -p $out
for rep in 1 2; do for ho in 0 1; do
OMNIHD_SPLIT_HANDOVER=$ho timeout 200 python3 scripts/lab/fault_repro.py fp32 40 > $out/step_$ho.txt 2> $out/step_$ho.err; echo "handover=$ho rc $? $(tail -1 $out/step_$ho.txt)"
done; done
