#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r4e; mkdir -p $out
timeout 600 python3 -m pytest tests/test_pfn_gpu.py -m gpu -x -q -k "two_ranks or scatter" 2>&1 | tail -4 > $out/pfn2.txt; cat $out/pfn2.txt
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 6000 $out/bench_default.json; tail -3 $out/bench_default.err
