#!/bin/bash
# Round-5 GPU call 27: forward plane handover on/off (A/B, two runs each).
export TMPDIR=/tmp; out=gpurun_out/r5aa; mkdir -p $out
for v in 0 1 0 1; do
  OMNIHD_SPLIT_HANDOVER=$v OMNIHD_BENCH_DDP1=0 python3 bench.py --dtype fp32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('handover $v', d['ms_per_step'], d['step_ms'])"
done | tee $out/handover_ab.txt
