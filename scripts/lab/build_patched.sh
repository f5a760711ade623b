#!/bin/bash
# Lab builds of libomnihd_hip.so from the PRODUCT sources with one patch applied (scripts/lab/patches/*.patch: ablation / trace
# hooks are kept as diffs against the product kernels, not as forked copies).
#   usage: build_patched.sh <patch-name> <macro> <value> [<value> ...]   ->  scripts/micro/abl/lib_<patch-name>_<value>.so
set -e
cd "$(dirname "$0")/../.."
ROOT=$PWD; NAME=$1; MACRO=$2; shift 2
SRC=$ROOT/scripts/micro/abl/src_$NAME
rm -rf $SRC; mkdir -p $SRC
for f in $ROOT/omnihd-scenes_amd/csrc/*.hip $ROOT/omnihd-scenes_amd/csrc/*.h $ROOT/omnihd-scenes_amd/csrc/Makefile; do cp $f $SRC/; done
if [ -f $ROOT/scripts/lab/patches/on_superseded/$NAME.patch ]; then   # hooks of kernels that left the product in round 6: restore those first
  patch -s $SRC/bev_pool_v2.hip < $ROOT/scripts/lab/patches/pool_superseded_kernels.patch
  patch -s $SRC/bev_pool_v2.hip < $ROOT/scripts/lab/patches/on_superseded/$NAME.patch
else
  patch -s $SRC/bev_pool_v2.hip < $ROOT/scripts/lab/patches/$NAME.patch   # (every patch so far is against bev_pool_v2.hip)
fi
for V in "$@"; do
  ( make -s -C $SRC -j3 ROOT=$ROOT OUTDIR=$ROOT/scripts/micro/abl/build_${NAME}_$V EXTRA=-D$MACRO=$V &&
    cp scripts/micro/abl/build_${NAME}_$V/libomnihd_hip.so scripts/micro/abl/lib_${NAME}_$V.so ) &
done
wait
ls -la scripts/micro/abl/lib_${NAME}_*.so
