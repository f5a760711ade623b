#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5final5; mkdir -p $out
timeout 1800 python3 -m pytest tests/test_bench_launch.py -m gpu -q 2>&1 | tail -4 > $out/tests.txt; cat $out/tests.txt
