#!/bin/bash
# Build the ablation variants of the library used by pool_traffic_abl.sh / pool_time_abl.sh (CPU box, hipcc cross-compiles).
# OMNIHD_POOL_ABL bits: 1 depth gather -> constant, 2 feature gathers -> 1024 L2-resident rows, 4 no pooled-row stores,
# 8 no zero-fill stores.  The product library is built WITHOUT the macro (make -C omnihd-scenes_amd/csrc).
set -e
cd "$(dirname "$0")/../omnihd-scenes_amd/csrc"
mkdir -p ../../scripts/micro/abl
for A in ${@:-1 2 3 4 8 12 15}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I../../include -DOMNIHD_POOL_ABL=$A \
    core.hip bev_pool_v2.hip bev_pool_v1.hip rank_prep.hip voxelize.hip pillar_scatter.hip conv_wgrad.hip dcn_sample.hip \
    nms_rotated.hip affine_act.hip batch_norm.hip -o ../../scripts/micro/abl/libomnihd_abl$A.so &
done
wait
ls -la ../../scripts/micro/abl/
