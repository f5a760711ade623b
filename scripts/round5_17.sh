#!/bin/bash
# Round-5 GPU call 17: split-count sweep of the three-taps weight gradient (how much of the chip does a launch keep busy?).
export TMPDIR=/tmp; out=gpurun_out/r5q; mkdir -p $out
for sp in 1 2 3 4 5 6 8; do
  for g in 1,160,240,1024,1024,3,1,1 6,64,176,256,256,3,1,1 1,160,240,512,256,3,1,1; do
    echo -n "splits $sp: "; OMNIHD_WGRAD_NHWC_SPLITS=$sp WGRAD_BENCH_ONE=$g WGRAD_BENCH_LIBRARY=0 WGRAD_BENCH_CHAIN=0 timeout 300 python3 scripts/lab/wgrad_nhwc_bench.py 2>&1 | grep "nhwc" | cut -c1-70
  done
done > $out/split_sweep.txt 2>&1; cat $out/split_sweep.txt
