#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5af; mkdir -p $out
timeout 900 python3 scripts/cpu_profile.py fp32 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/cpu_profile_fp32.txt; head -48 $out/cpu_profile_fp32.txt | cut -c1-170
