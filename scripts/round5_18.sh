#!/bin/bash
# Round-5 GPU call 18: counters of the three-taps weight gradient at 192 workgroups (one split) vs the default, and on 512->256.
export TMPDIR=/tmp; out=gpurun_out/r5r; mkdir -p $out
OMNIHD_WGRAD_NHWC_SPLITS=1 bash scripts/lab/pmc_wgrad.sh $out/pmc_1024_s1 1,160,240,1024,1024,3,1,1 > $out/pmc_wgrad_1024_s1.txt 2>&1; head -22 $out/pmc_wgrad_1024_s1.txt
bash scripts/lab/pmc_wgrad.sh $out/pmc_512 1,160,240,512,256,3,1,1 > $out/pmc_wgrad_512.txt 2>&1; head -22 $out/pmc_wgrad_512.txt
OMNIHD_WGRAD_NHWC_SPLITS=2 bash scripts/lab/pmc_wgrad.sh $out/pmc_512_s2 1,160,240,512,256,3,1,1 > $out/pmc_wgrad_512_s2.txt 2>&1; head -22 $out/pmc_wgrad_512_s2.txt
