#!/bin/bash
export TMPDIR=/tmp
for sw in NONE=1 OMNIHD_DECONV_SPLIT=0 OMNIHD_CONV_GEN=0 OMNIHD_WGRAD_NHWC=0 OMNIHD_SPLIT_HANDOVER=0 OMNIHD_GRAD_PLANES_ONLY=0 OMNIHD_WGRAD_OVERLAP=0 OMNIHD_BN_MASK_FROM_X=0; do
echo "== split + $sw"; env $sw OMNIHD_FP32_CONV=split OMNIHD_POOL_DEVICE_PLAN=0 timeout 900 python3 scripts/lab/occ_flaky.py 1 2>&1 | grep "^run" | cut -c60-600
done
