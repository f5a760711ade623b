#!/bin/bash
# Round-4 first GPU call: counters of the pooling kernels at R2 (the repo's own resolution) + gather microbenchmark + GPU suite.
export TMPDIR=/tmp; out=gpurun_out/r4a; mkdir -p $out
bash scripts/lab/pmc_bwd.sh $out/pmc_r2 r2 > $out/pmc_r2.log 2>&1
tail -60 $out/pmc_r2.log
python3 scripts/sweep_lean.py r2 2>&1 | grep -v "^/opt" | tail -12 > $out/sweep_r2.txt; cat $out/sweep_r2.txt
(cd scripts/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 gather_bw.hip -o gather_bw && ./gather_bw) > $out/gather_bw.txt 2>&1; tail -40 $out/gather_bw.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $out/gputests.txt; cat $out/gputests.txt
