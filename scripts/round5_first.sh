#!/bin/bash
# Round-5 first GPU call: the tests touched so far (ADVICE fixes, DDP bucket hook), then the default bench line.
export TMPDIR=/tmp; out=gpurun_out/r5a; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_conv_split_gpu.py tests/test_radar_gpu.py tests/test_bench_launch.py tests/test_ddp_shared_gpu.py \
  "tests/test_detector_gpu.py::test_ddp_wrapped_step_on_one_gpu" -m gpu -x -q 2>&1 | tail -25 > $out/tests.txt; cat $out/tests.txt
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cat $out/bench.json
