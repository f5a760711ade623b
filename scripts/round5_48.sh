#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5ao; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_conv_split_gpu.py tests/test_bench_launch.py tests/test_ddp_shared_gpu.py -m gpu -q 2>&1 | tail -3 > $out/tests.txt; cat $out/tests.txt
run() { echo "== $MODE $*"; env "$@" python3 scripts/lab/ddp1_step.py $MODE 2>&1 | grep "ms/step"; }
{ MODE=plain run A=1; MODE=ddp run A=1; MODE=plain run A=1; MODE=ddp run A=1; } | tee $out/ddp_hook_skip.txt
