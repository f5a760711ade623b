#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r4d; mkdir -p $out
timeout 900 python3 -m pytest tests/test_pfn_gpu.py tests/test_radar_gpu.py tests/test_pillars_gpu.py -m gpu -x -q 2>&1 | tail -25 > $out/pfn_tests.txt; cat $out/pfn_tests.txt
OMNIHD_CHOICE_TABLE=off timeout 1500 python3 scripts/capture_choice_table.py $out/gfx950.json 2>&1 | grep -v "^/opt\|Warning\|warn" | tail -8
