#!/bin/bash
# Round 6 (last GPU seconds): the frozen-BatchNorm epilogue's hi / lo hand-over (opt-in) and the half hand-over that shares its code
export TMPDIR=/tmp; out=gpurun_out/r6_30; mkdir -p $out
timeout 95 python3 -m pytest tests/test_conv_split_gpu.py tests/test_conv_f16_gpu.py -q -p no:cacheprovider -x -k "frozen_batchnorm_epilogue or producers_hand_over or batchnorm_hands" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest.txt
