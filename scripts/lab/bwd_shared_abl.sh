#!/bin/bash
# Where the time of k_pool_bwd_shared goes: phase timeline (ABL=1) and launch time with the point loop (2), the row gathers (4)
# or both (6) compiled out.  usage (GPU box): bash scripts/lab/bwd_shared_abl.sh [pw] [R]
cd "$(dirname "$0")/../.."
PW=${1:-8}; R=${2:-64}
for RES in r1 r2; do
  OMNIHD_LIB_PATH=$PWD/scripts/micro/abl/lib_pool_bwd_shared_instrument_1.so timeout 200 python3 scripts/lab/bwd_shared_trace.py $RES $PW $R 2>&1 | grep -v amdgpu.ids
  for V in 2 4 6; do
    echo "--- ABL=$V (2: no point loop, 4: no row gathers, 6: neither)"
    OMNIHD_LIB_PATH=$PWD/scripts/micro/abl/lib_pool_bwd_shared_instrument_$V.so timeout 200 python3 scripts/lab/ab_bwd_shared.py $RES --shapes $PW --rows $R 2>&1 | grep "shared"
  done
done
