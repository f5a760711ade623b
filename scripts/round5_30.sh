#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5ad; mkdir -p $out
timeout 900 python3 scripts/lab/tiny_grads_split.py 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/tiny_grads_split.txt; cat $out/tiny_grads_split.txt | cut -c1-150
for pol in tune hip; do
  OMNIHD_CONV_POLICY=$pol OMNIHD_WGRAD_POLICY=$pol OMNIHD_BENCH_DDP1=0 python3 bench.py --dtype bf16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 policy $pol', d['ms_per_step'], d['step_ms'])"
done | tee $out/bf16_policy.txt
