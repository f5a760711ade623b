#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5al; mkdir -p $out
run() { echo "== $*"; env "$@" python3 scripts/lab/ddp1_step.py $MODE 2>&1 | grep "ms/step"; }
{ MODE=plain run A=1; MODE=ddp run A=1; MODE=ddp run OMNIHD_DDP_BUCKET_MB=100; MODE=ddp run OMNIHD_DDP_STATIC=1; MODE=ddp run OMNIHD_DDP_BUCKET_MB=100 OMNIHD_DDP_STATIC=1; MODE=ddp run OMNIHD_DDP_HOOK=0; MODE=plain run A=1; MODE=ddp run A=1; } | tee $out/ddp_variants2.txt
