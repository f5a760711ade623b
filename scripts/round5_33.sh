#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5ag; mkdir -p $out
timeout 900 python3 scripts/trace_copies.py fp32 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/copies_fp32.txt; head -50 $out/copies_fp32.txt | cut -c1-260
