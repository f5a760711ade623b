#!/bin/bash
# Timing ablation of the row-shift convolution kernel (split and bf16 forms): which part of a K-step costs what.
# Builds the library with scripts/lab/patches/conv_igemm_abl.patch applied to csrc/conv_igemm.hip for each OMNIHD_CONV_ABL value
# (CPU box, hipcc cross-compiles), then `bash scripts/lab/conv_abl.sh run` on the GPU box times 1024->1024 @160x240.
set -e
cd "$(dirname "$0")/../.."
ROOT=$PWD
if [ "$1" = run ]; then
  for A in ${@:2}; do
    echo "== OMNIHD_CONV_ABL=$A"
    OMNIHD_LIB_PATH=$ROOT/scripts/micro/convabl/libomnihd_convabl$A.so python3 scripts/lab/conv_abl_time.py 2>&1 | grep -v "^/opt"
  done
  exit 0
fi
SRC=$ROOT/scripts/micro/convabl/src
mkdir -p $SRC
for f in $ROOT/omnihd-scenes_amd/csrc/*.hip $ROOT/omnihd-scenes_amd/csrc/*.h $ROOT/omnihd-scenes_amd/csrc/Makefile; do ln -sf $f $SRC/; done
rm -f $SRC/conv_igemm.hip; cp $ROOT/omnihd-scenes_amd/csrc/conv_igemm.hip $SRC/conv_igemm.hip
patch -s $SRC/conv_igemm.hip < $ROOT/scripts/lab/patches/conv_igemm_abl.patch
for A in ${@:-0 1 2 4 8 7 14}; do
  # A = <abl bits>[g<group rule>]  e.g. 0g1 = no ablation, groups by wave parity
  ( make -s -C $SRC -j2 ROOT=$ROOT OUTDIR=$ROOT/scripts/micro/convabl/build$A EXTRA="-DOMNIHD_CONV_ABL=${A%%g*} -DOMNIHD_CONV_GRP=$(echo $A | sed -n 's/.*g//p' | grep . || echo 0)" &&
    cp scripts/micro/convabl/build$A/libomnihd_hip.so scripts/micro/convabl/libomnihd_convabl$A.so && rm -rf scripts/micro/convabl/build$A ) &
done
wait
ls -la scripts/micro/convabl/*.so
