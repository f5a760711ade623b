#!/bin/bash
# Round-5 GPU call 14: NHWC weight gradient with the XCD-aware work order (parity, timing), chain tests in both forms.
export TMPDIR=/tmp; out=gpurun_out/r5n; mkdir -p $out
timeout 900 python3 -m pytest tests/test_wgrad_nhwc_gpu.py -m gpu -q -x 2>&1 | tail -25 > $out/wgrad_nhwc_tests.txt; cat $out/wgrad_nhwc_tests.txt
timeout 900 python3 scripts/lab/wgrad_nhwc_bench.py 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/wgrad_nhwc_bench.txt; cat $out/wgrad_nhwc_bench.txt
timeout 1200 python3 -m pytest tests/test_conv_gpu.py tests/test_conv_split_gpu.py -m gpu -q -x -k "wgrad or weight_gradient" 2>&1 | tail -8 > $out/chain_tests.txt; cat $out/chain_tests.txt
