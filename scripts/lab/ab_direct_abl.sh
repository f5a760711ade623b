#!/bin/bash
# direct forward kernel: ablation builds (scripts/lab/patches/pool_direct_abl.patch) timed at a resolution
RES=${1:-r2}
python3 scripts/lab/ab_direct.py $RES 2>&1 | grep "direct keep_zeros=1" | sed "s/^/product /"
for A in 1 2 4 16 20 53; do
  OMNIHD_LIB_PATH=$PWD/scripts/micro/abl/lib_pool_direct_abl_$A.so python3 scripts/lab/ab_direct.py $RES 2>&1 | grep "direct keep_zeros=1" | sed "s/^/abl $A /"
done
