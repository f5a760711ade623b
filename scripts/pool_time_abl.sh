#!/bin/bash
# Time skeletons of the one-table pooling forward (ablation builds in scripts/micro/abl/, see csrc/bev_pool_v2.hip).
python3 scripts/ab_lean.py r1 2>&1 | grep "rep 2" | sed 's/^/full      /'
for A in 1 2 3 4 8 12 15; do
  OMNIHD_LIB_PATH=$PWD/scripts/micro/abl/libomnihd_abl$A.so python3 scripts/ab_lean.py r1 2>&1 | grep "rep 2" | sed "s/^/abl $A     /"
done
