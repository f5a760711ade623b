#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r6_05b; mkdir -p $out
for i in 1 2 3; do
  OMNIHD_POOL_DEVICE_PLAN=0 timeout 600 python3 -m pytest tests/test_detector_gpu.py -q -k "occupancy_variant" > $out/occ_host_$i.txt 2>&1; echo "host plan run $i rc $?"
  timeout 600 python3 -m pytest tests/test_detector_gpu.py -q -k "occupancy_variant" > $out/occ_dev_$i.txt 2>&1; echo "device plan run $i rc $?"
done
