#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r6_06; mkdir -p $out
timeout 900 python3 scripts/lab/plan_stress.py 150 256 704 410 2>&1 | tail -12
timeout 900 python3 scripts/lab/plan_stress.py 400 64 96 60 2>&1 | tail -12
timeout 900 python3 scripts/lab/plan_stress.py 60 544 960 560 2>&1 | tail -12
