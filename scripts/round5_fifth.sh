#!/bin/bash
# Round-5 fifth GPU call: fused anchor loss + grid voxelisation tests, default bench line, deterministic-policy step time.
export TMPDIR=/tmp; out=gpurun_out/r5e; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_anchor_loss_gpu.py tests/test_radar_gpu.py tests/test_radar_properties_gpu.py tests/test_conv_split_gpu.py tests/test_pillars_gpu.py tests/test_detector_gpu.py -m gpu -q 2>&1 | tail -30 > $out/tests.txt; cat $out/tests.txt
timeout 900 python3 bench.py > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.err; cat $out/bench.json
OMNIHD_DETERMINISTIC=1 timeout 600 python3 bench.py --dtype fp32 --no-cpu-baseline > $out/bench_deterministic.json 2> $out/bench_deterministic.err; cat $out/bench_deterministic.json | cut -c1-400
