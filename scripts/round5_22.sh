#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5v; mkdir -p $out
timeout 900 python3 scripts/lab/split_census.py 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/split_census.txt; head -70 $out/split_census.txt | cut -c1-200
