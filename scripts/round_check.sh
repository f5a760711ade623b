#!/bin/bash
# Round-end style check on the GPU box: GPU tests, smoke, bench line, rocprofv3 kernel stats of the bench command.
export TMPDIR=/tmp; mkdir -p gpurun_out/round
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/round/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/round/smoke.txt 2>&1
python3 bench.py > gpurun_out/round/bench.json 2> gpurun_out/round/bench.err
rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/round/prof -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/round/bench_prof.json 2> gpurun_out/round/bench_prof.err
cp $(find gpurun_out/round/prof -name "*kernel_stats.csv" | head -1) gpurun_out/round/bench_kernel_stats.csv
find gpurun_out/round/prof -type f -size +2M -delete
tail -2 gpurun_out/round/pytest_gpu.txt; tail -1 gpurun_out/round/smoke.txt; cat gpurun_out/round/bench.json; tail -1 gpurun_out/round/bench_prof.json
