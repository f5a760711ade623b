#!/bin/bash
# Round-5 GPU call 23: host slack of the fp32 / bf16 step; kernel-choice table recapture.
export TMPDIR=/tmp; out=gpurun_out/r5w; mkdir -p $out
timeout 600 python3 scripts/lab/host_slack.py fp32 2>&1 | grep -v "^/opt\|Warn\|warn" > $out/host_slack.txt
timeout 600 python3 scripts/lab/host_slack.py bf16 2>&1 | grep -v "^/opt\|Warn\|warn" >> $out/host_slack.txt; cat $out/host_slack.txt
OMNIHD_CHOICE_TABLE=off timeout 2400 python3 scripts/capture_choice_table.py $out/gfx950.json 2>&1 | grep -v "^/opt\|Warn\|warn" | tail -8 > $out/capture.txt; cat $out/capture.txt
