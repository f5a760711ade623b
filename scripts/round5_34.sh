#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5ah; mkdir -p $out
for v in 0 1 0 1; do
  OMNIHD_RESIZE_CL=$v OMNIHD_BENCH_DDP1=0 python3 bench.py --dtype fp32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resize_cl $v', d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10'])"
done | tee $out/resize_cl_ab.txt
