#!/bin/bash
# Round-5 seventh GPU call: which passes stay on the library under the deterministic policy; determinism without the cudnn flag;
# planes-only gradients (tests + step time); early radar join A/B.
export TMPDIR=/tmp; out=gpurun_out/r5g; mkdir -p $out
timeout 600 python3 scripts/lab/det_leftovers.py 2>&1 | grep -v "^/opt\|Warn\|warn" | tail -40 > $out/det_leftovers.txt; cat $out/det_leftovers.txt
OMNIHD_DETERMINISTIC=1 OMNIHD_DET_CUDNN=0 timeout 900 python3 scripts/lab/determinism_pass.py 2>&1 | grep -v "^/opt\|Warn\|warn" | head -30 > $out/determinism_nocudnnflag.txt; cat $out/determinism_nocudnnflag.txt
timeout 1500 python3 -m pytest tests/test_conv_split_gpu.py tests/test_bn_gpu.py tests/test_stage_gradients_gpu.py tests/test_detector_gpu.py tests/test_lss_plain_gpu.py -m gpu -q -x 2>&1 | tail -15 > $out/tests.txt; cat $out/tests.txt
for v in "OMNIHD_GRAD_PLANES_ONLY=1" "OMNIHD_GRAD_PLANES_ONLY=0" "OMNIHD_RADAR_JOIN=early" "OMNIHD_GRAD_PLANES_ONLY=1"; do
  echo "== $v" >> $out/variants.txt
  env $v timeout 300 python3 scripts/lab/ddp1_step.py plain 2>&1 | grep "ms/step" >> $out/variants.txt
done
cat $out/variants.txt
