#!/bin/bash
# Build the ablation / trace variants of the library used by pool_traffic_abl.sh / pool_time_abl.sh / lab/pool_trace.py (CPU box,
# hipcc cross-compiles).  The instrumentation (OMNIHD_POOL_ABL / OMNIHD_POOL_TRACE hooks of the lean forward kernels and the patch
# backward) is kept as a PATCH against the product source, scripts/lab/patches/on_superseded/pool_lean_instrument.patch; the product source
# omnihd-scenes_amd/csrc/bev_pool_v2.hip carries none.  Every variant is a full build of the product Makefile's source list with
# that one file patched, into its own directory.
# OMNIHD_POOL_ABL bits: 1 depth gather -> constant, 2 feature gathers -> 1024 L2-resident rows, 4 no pooled-row stores,
# 8 no zero-fill stores, 16 no feature gathers at all.  `build_abl.sh trace` builds the -DOMNIHD_POOL_TRACE library.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
SRC=$ROOT/scripts/micro/abl/src
mkdir -p $SRC
for f in $ROOT/omnihd-scenes_amd/csrc/*.hip $ROOT/omnihd-scenes_amd/csrc/*.h $ROOT/omnihd-scenes_amd/csrc/Makefile; do ln -sf $f $SRC/; done
rm -f $SRC/bev_pool_v2.hip; cp $ROOT/omnihd-scenes_amd/csrc/bev_pool_v2.hip $SRC/bev_pool_v2.hip
# (round 6: the LDS-staged forward kernels left the product; their instrumentation applies on top of the patch that restores them)
patch -s $SRC/bev_pool_v2.hip < $ROOT/scripts/lab/patches/pool_superseded_kernels.patch
patch -s $SRC/bev_pool_v2.hip < $ROOT/scripts/lab/patches/on_superseded/pool_lean_instrument.patch
for A in ${@:-1 2 3 4 8 12 15 16 28}; do
  if [ "$A" = trace ]; then FLAG=-DOMNIHD_POOL_TRACE; OUT=libomnihd_trace.so; else FLAG=-DOMNIHD_POOL_ABL=$A; OUT=libomnihd_abl$A.so; fi
  ( make -s -C $SRC -j2 ROOT=$ROOT OUTDIR=$ROOT/scripts/micro/abl/build$A EXTRA=$FLAG &&
    cp scripts/micro/abl/build$A/libomnihd_hip.so scripts/micro/abl/$OUT ) &
done
wait
ls -la scripts/micro/abl/*.so
