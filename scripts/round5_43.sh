#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5ak; mkdir -p $out
for i in 1 2; do
  python3 bench.py --dtype fp32 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i plain', d['ms_per_step'], d['step_ms']['median'], '| ddp', d['ddp_1rank']['ms_per_step'], d['ddp_1rank']['step_ms'], d['ddp_1rank']['overhead_vs_plain'])"
done | tee $out/ddp_after_setup.txt
