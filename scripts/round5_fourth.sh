#!/bin/bash
# Round-5 fourth GPU call: re-capture the kernel-choice table (new general-kernel geometries), one-rank DDP overhead by variant, GPU suite.
export TMPDIR=/tmp; out=gpurun_out/r5d; mkdir -p $out
timeout 1500 python3 scripts/capture_choice_table.py $out/gfx950.json 2>&1 | grep -v "^/opt\|Warn\|warn" | tail -8 > $out/capture.txt; cat $out/capture.txt
[ -s $out/gfx950.json ] && cp $out/gfx950.json omnihd-scenes_amd/kernel_choices/gfx950.json
for v in "plain" "ddp" "plain OMNIHD_DUAL_STREAM=0" "ddp OMNIHD_DUAL_STREAM=0" "ddp OMNIHD_DDP_SMALL_FLAT=0" "ddp OMNIHD_WGRAD_OVERLAP=0" "plain OMNIHD_WGRAD_OVERLAP=0"; do
  set -- $v; mode=$1; shift
  echo "== $v" >> $out/ddp_variants.txt
  env "$@" MASTER_PORT=$((29800 + RANDOM % 100)) timeout 300 python3 scripts/lab/ddp1_step.py $mode 2>&1 | grep "ms/step\|hooked" >> $out/ddp_variants.txt
done
cat $out/ddp_variants.txt
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -25 > $out/gputests.txt; cat $out/gputests.txt
