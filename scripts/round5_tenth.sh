#!/bin/bash
# Round-5 tenth GPU call: NHWC weight gradient (parity, per-geometry timing), determinism test with diagnostics.
export TMPDIR=/tmp; out=gpurun_out/r5j; mkdir -p $out
timeout 900 python3 -m pytest tests/test_wgrad_nhwc_gpu.py -m gpu -q -x 2>&1 | tail -25 > $out/wgrad_nhwc_tests.txt; cat $out/wgrad_nhwc_tests.txt
timeout 900 python3 scripts/lab/wgrad_nhwc_bench.py 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/wgrad_nhwc_bench.txt; cat $out/wgrad_nhwc_bench.txt
timeout 900 python3 -m pytest tests/test_determinism_gpu.py -m gpu -q 2>&1 | tail -12 | cut -c1-1500 > $out/det_test.txt; cat $out/det_test.txt
