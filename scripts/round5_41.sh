#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5ai; mkdir -p $out
for st in 2 3; do
  echo "--- one-tap stages $st"
  OMNIHD_WGRAD_NHWC_STAGES=$st WGRAD_BENCH_LIBRARY=0 WGRAD_BENCH_CHAIN=0 timeout 600 python3 scripts/lab/wgrad_nhwc_bench.py 2>&1 | grep "k1\|s2" | cut -c1-75
done | tee $out/onetap_stages.txt
