#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5ac; mkdir -p $out
timeout 900 python3 scripts/cpu_profile.py 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/cpu_profile_bf16.txt; head -60 $out/cpu_profile_bf16.txt | cut -c1-170
