#!/bin/bash
# Capture MIOpen's user database (find-db / perf-db records + compiled kernels) for the workloads the bench and the GPU tests
# run, starting from the database kept in omnihd-scenes_amd/miopen_db: run on the GPU box, then copy gpurun_out/miopen_db_new/*
# over omnihd-scenes_amd/miopen_db/ (the .ukdb is git-ignored but travels with the working tree; the text dbs are tracked).
set -x
export TMPDIR=/tmp
out=$PWD/gpurun_out/miopen_db_new; mkdir -p $out; cp omnihd-scenes_amd/miopen_db/* $out/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$out MIOPEN_CUSTOM_CACHE_DIR=$out
( time python -m pytest tests/test_triple_gpu.py -m gpu -x -q -s -k full_size 2>&1 | tail -6 ) > gpurun_out/triple_full.txt 2>&1
python -m pytest tests/test_detector_gpu.py -m gpu -x -q -k "camera_only or full_size" 2>&1 | tail -4 > gpurun_out/cap_tests.txt
python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/cap_bench.json 2> gpurun_out/cap_bench.err
python scripts/infer_fps.py r1 bf16 100 det 2>&1 | tail -2 > gpurun_out/infer_fps.txt
python scripts/infer_fps.py r1 bf16 100 camera 2>&1 | tail -2 >> gpurun_out/infer_fps.txt
python scripts/infer_fps.py r1 fp32 60 det 2>&1 | tail -1 >> gpurun_out/infer_fps.txt
python scripts/infer_fps.py r1 fp32 60 camera 2>&1 | tail -1 >> gpurun_out/infer_fps.txt
du -sh $out; ls -la $out
cat gpurun_out/triple_full.txt gpurun_out/cap_tests.txt gpurun_out/infer_fps.txt
