#!/bin/bash
# Round-5 GPU call 11: cross-process determinism diagnosis.
export TMPDIR=/tmp; out=gpurun_out/r5k; mkdir -p $out
timeout 1500 python3 scripts/lab/det_cross.py 2 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/det_cross.txt; cat $out/det_cross.txt
