#!/bin/bash
# Round-5 GPU call 13: NHWC weight gradient with the three-taps form (parity, timing with / without it), determinism under a seeded db.
export TMPDIR=/tmp; out=gpurun_out/r5m; mkdir -p $out
timeout 900 python3 -m pytest tests/test_wgrad_nhwc_gpu.py -m gpu -q -x 2>&1 | tail -25 > $out/wgrad_nhwc_tests.txt; cat $out/wgrad_nhwc_tests.txt
timeout 900 python3 scripts/lab/wgrad_nhwc_bench.py 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/wgrad_nhwc_bench.txt; cat $out/wgrad_nhwc_bench.txt
echo "--- one tap only (OMNIHD_WGRAD_NHWC_THREE=0), 3x3 layers"
WGRAD_BENCH_3X3_ONLY=1 WGRAD_BENCH_LIBRARY=0 OMNIHD_WGRAD_NHWC_THREE=0 timeout 900 python3 scripts/lab/wgrad_nhwc_bench.py 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/wgrad_nhwc_bench_onetap.txt; cat $out/wgrad_nhwc_bench_onetap.txt
echo "--- one tap, 3 stages"
WGRAD_BENCH_LIBRARY=0 OMNIHD_WGRAD_NHWC_THREE=0 OMNIHD_WGRAD_NHWC_STAGES=3 timeout 900 python3 scripts/lab/wgrad_nhwc_bench.py 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/wgrad_nhwc_bench_onetap3.txt; cat $out/wgrad_nhwc_bench_onetap3.txt
timeout 900 python3 -m pytest tests/test_determinism_gpu.py -m gpu -q -x 2>&1 | grep -v "^/opt\|Warn\|warn" | cut -c1-3000 | tail -12 > $out/det_test.txt; tail -5 $out/det_test.txt | cut -c1-400
