#!/bin/bash
# Round-5 GPU call 19: persistent NHWC weight gradient with the pinned fill schedule (parity, timing, counters).
export TMPDIR=/tmp; out=gpurun_out/r5s; mkdir -p $out
timeout 900 python3 -m pytest tests/test_wgrad_nhwc_gpu.py -m gpu -q -x 2>&1 | tail -25 > $out/wgrad_nhwc_tests.txt; cat $out/wgrad_nhwc_tests.txt
WGRAD_BENCH_LIBRARY=0 WGRAD_BENCH_CHAIN=0 timeout 900 python3 scripts/lab/wgrad_nhwc_bench.py 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/wgrad_nhwc_bench.txt; cat $out/wgrad_nhwc_bench.txt
bash scripts/lab/pmc_wgrad.sh $out/pmc_1024 1,160,240,1024,1024,3,1,1 > $out/pmc_wgrad_1024.txt 2>&1; head -22 $out/pmc_wgrad_1024.txt
