#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5ab; mkdir -p $out
for i in 1 2 3; do
  OMNIHD_BENCH_DDP1=0 python3 bench.py --dtype fp32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i', d['ms_per_step'], d['step_ms'])"
done | tee $out/slow_steps.txt
