#!/bin/bash
# Round 6, first GPU call: the device-built pooling plan — parity tests, build time, per-kernel breakdown.
export TMPDIR=/tmp; out=gpurun_out/r6_01; mkdir -p $out
timeout 900 python3 -m pytest tests/test_device_plan_gpu.py -x -q > $out/pytest_device_plan.txt 2>&1; echo "pytest rc $?"; tail -25 $out/pytest_device_plan.txt
timeout 600 python3 scripts/round6_plan.py both > $out/plan_times.jsonl 2> $out/plan_times.err; echo "plan rc $?"; cat $out/plan_times.jsonl; tail -5 $out/plan_times.err
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof -o plan -- python3 scripts/round6_plan.py r1 --iters 10 > $out/plan_prof.json 2> $out/plan_prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/plan_kernel_stats_r1.csv 2>/dev/null
head -40 $out/plan_kernel_stats_r1.csv | cut -c1-220
find $out/prof -type f -size +2M -delete
