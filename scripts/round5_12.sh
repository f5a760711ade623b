#!/bin/bash
# Round-5 GPU call 12: the determinism test as pytest runs it (full diff list), twice; then the lab variant list.
export TMPDIR=/tmp; out=gpurun_out/r5l; mkdir -p $out
for i in 1 2; do
  timeout 900 python3 -m pytest tests/test_determinism_gpu.py -m gpu -q -x 2>&1 | grep -v "^/opt\|Warn\|warn" | cut -c1-6000 | tail -40 > $out/det_test_$i.txt; tail -5 $out/det_test_$i.txt | cut -c1-300
done
DET_VARIANTS=default,in timeout 900 python3 scripts/lab/det_cross.py 2 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/det_cross.txt; cat $out/det_cross.txt
