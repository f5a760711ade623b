#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r4b; mkdir -p $out
python3 scripts/lab/pool_r2_probe.py r2 2>&1 | grep -v "^/opt" > $out/probe_product.txt
for A in 1 2 16 12; do OMNIHD_LIB_PATH=$PWD/scripts/micro/abl/libomnihd_abl$A.so python3 scripts/lab/pool_r2_probe.py r2 2>&1 | grep -v "^/opt" > $out/probe_abl$A.txt; done
OMNIHD_LIB_PATH=$PWD/scripts/micro/abl/libomnihd_trace.so python3 scripts/lab/pool_trace.py r2 2>&1 | grep -v "^/opt" > $out/trace_r2.txt
cat $out/probe_*.txt $out/trace_r2.txt
