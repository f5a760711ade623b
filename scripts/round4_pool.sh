#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r4c; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_bev_pool_gpu.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -5 > $out/pooltests.txt; cat $out/pooltests.txt
timeout 1500 python3 -m pytest tests/test_detector_gpu.py -m gpu -x -q -k "full_size" 2>&1 | tail -5 > $out/dettests.txt; cat $out/dettests.txt
bash scripts/lab/pmc_bwd.sh $out/pmc_r2 r2 > $out/pmc_r2.log 2>&1; tail -3 $out/pmc_r2.log
bash scripts/lab/pmc_bwd.sh $out/pmc_r1 r1 > $out/pmc_r1.log 2>&1; tail -3 $out/pmc_r1.log
