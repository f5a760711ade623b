#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5aj; mkdir -p $out
timeout 900 python3 scripts/lab/cpu_profile_ddp.py 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/cpu_profile_ddp.txt; head -60 $out/cpu_profile_ddp.txt | cut -c1-170
