#!/bin/bash
# Where the time of k_pool_bwd_stream goes: iteration timeline (ABL=1) and launch time with the point loop (2), the row gathers (4),
# both (6), or both + the per-patch pixel data and stores (14) compiled out.  usage (GPU box): bash scripts/lab/bwd_stream_abl.sh [cfg pw:R:streams]
cd "$(dirname "$0")/../.."
CFG=${1:-8:32:256}
IFS=: read PW R SPX <<< "$CFG"
for RES in r1 r2; do
  OMNIHD_LIB_PATH=$PWD/scripts/micro/abl/lib_pool_bwd_stream_instrument_1.so timeout 200 python3 scripts/lab/bwd_stream_trace.py $RES $PW $R $SPX 2>&1 | grep -v amdgpu.ids
  for V in 2 4 6 14; do
    echo "--- ABL=$V (2: no point loop, 4: no row gathers, 8: no pixel data / gradient stores)"
    OMNIHD_LIB_PATH=$PWD/scripts/micro/abl/lib_pool_bwd_stream_instrument_$V.so timeout 200 python3 scripts/lab/ab_bwd_stream.py $RES --stream $CFG 2>&1 | grep "stream" | sed 's/| feat_grad.*//'
  done
done
