#!/bin/bash
# Round-5 eighth GPU call: planes-only gradients (tests + step time), early radar join, deterministic step without the cudnn flag.
export TMPDIR=/tmp; out=gpurun_out/r5h; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_conv_split_gpu.py tests/test_bn_gpu.py tests/test_stage_gradients_gpu.py tests/test_detector_gpu.py tests/test_lss_plain_gpu.py tests/test_determinism_gpu.py -m gpu -q 2>&1 | tail -15 > $out/tests.txt; cat $out/tests.txt
for v in "OMNIHD_GRAD_PLANES_ONLY=1" "OMNIHD_GRAD_PLANES_ONLY=0" "OMNIHD_RADAR_JOIN=early" "OMNIHD_GRAD_PLANES_ONLY=1" "OMNIHD_GRAD_PLANES_ONLY=0" "OMNIHD_DETERMINISTIC=1"; do
  echo "== $v" >> $out/variants.txt
  env $v timeout 300 python3 scripts/lab/ddp1_step.py plain 2>&1 | grep "ms/step" >> $out/variants.txt
done
cat $out/variants.txt
OMNIHD_DETERMINISTIC=1 timeout 900 python3 scripts/lab/determinism_pass.py 2>&1 | grep -v "^/opt\|Warn\|warn" | head -8 > $out/determinism_pass.txt; cat $out/determinism_pass.txt
