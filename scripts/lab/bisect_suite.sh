#!/bin/bash
# Which earlier GPU test file changes the arithmetic of the plain-LSS golden test?  (process-global state hunt)
out=gpurun_out/r2d; mkdir -p $out
T=tests/test_lss_plain_gpu.py::test_reference_forward_and_backward_through_the_hip_path
for f in test_bev_pool_gpu test_bn_gpu test_conv_gpu test_detector_gpu; do
  timeout 400 python -m pytest tests/$f.py $T -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|where False = _close" | cut -c1-200 | sed "s/^/$f: /" >> $out/bisect_suite.log
done
MIOPEN_ENABLE_LOGGING_CMD=1 MIOPEN_LOG_LEVEL=5 python scripts/repro_lss_grad.py test 2>&1 | grep -iE "solver|algorithm|MIOpenDriver|^test " | cut -c1-300 | head -n 200 > $out/miopen_default.log
MIOPEN_ENABLE_LOGGING_CMD=1 MIOPEN_LOG_LEVEL=5 python scripts/repro_lss_grad.py notf32 2>&1 | grep -iE "solver|algorithm|MIOpenDriver|^notf32 " | cut -c1-300 | head -n 200 > $out/miopen_notf32.log
cat $out/bisect_suite.log
