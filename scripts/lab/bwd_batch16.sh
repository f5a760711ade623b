#!/bin/bash
# Is k_pool_bwd_patch short of requests in flight?  Lab builds with all 16 row gathers of a chunk issued at once (instead of 8 + 8)
# at 4 / 5 / 6 waves per SIMD (scripts/lab/patches/pool_bwd_batch16.patch).  Result (profiles/round4/pool_bwd_batch16.txt): no change.
#   bash scripts/lab/build_patched.sh pool_bwd_batch16 OMNIHD_BWD_WAVES 4 5 6     (CPU box), then on the GPU box:
cd "$(dirname "$0")/../.."
for V in 4 5 6; do
  echo "--- 16 gathers in flight, $V waves per SIMD"
  OMNIHD_LIB_PATH=$PWD/scripts/micro/abl/lib_pool_bwd_batch16_$V.so timeout 200 python3 scripts/lab/ab_bwd_stream.py r1 r2 --stream "" 2>&1 | grep patch
done
