#!/bin/bash
# Build the ablation variants of the library used by pool_traffic_abl.sh / pool_time_abl.sh (CPU box, hipcc cross-compiles).
# OMNIHD_POOL_ABL bits: 1 depth gather -> constant, 2 feature gathers -> 1024 L2-resident rows, 4 no pooled-row stores,
# 8 no zero-fill stores, 16 no feature gathers at all.  The product library is built WITHOUT the macro.
# The source list is the product Makefile's (one list, cannot drift): each variant is a full build into its own directory.
set -e
cd "$(dirname "$0")/.."
mkdir -p scripts/micro/abl
for A in ${@:-1 2 3 4 8 12 15 16 28}; do
  ( make -s -C omnihd-scenes_amd/csrc -j2 OUTDIR=$PWD/scripts/micro/abl/build$A EXTRA=-DOMNIHD_POOL_ABL=$A &&
    cp scripts/micro/abl/build$A/libomnihd_hip.so scripts/micro/abl/libomnihd_abl$A.so ) &
done
wait
ls -la scripts/micro/abl/*.so
