"""omnihd_amd.mm — the slice of the mmcv / mmdet / mmdet3d surface that the reference's fusion
configs touch, restated on plain PyTorch so that the reference configs
(projects/configs/bevfusion_NewScenes/*.py) build WITHOUT those packages.

Everything here is host-side structure (registries, layer builders, module containers); the
arithmetic runs in torch/MIOpen ops or in the hand-written HIP kernels of libomnihd_hip.so.
Upstream packages are not vendored in the reference (README.md:143-166 pins mmcv-full 1.4.0,
mmdet 2.14.0, mmdet3d v0.17.1); module/attribute names follow those releases so that reference
checkpoints keep their state-dict keys.
"""
from .registry import (BACKBONES, DETECTORS, HEADS, LOSSES, MIDDLE_ENCODERS, NECKS, NORM_LAYERS,  # noqa: F401
                       PIPELINES, VOXEL_ENCODERS, Registry, build_from_cfg)
from .bricks import ConvModule, build_conv_layer, build_norm_layer, build_activation  # noqa: F401
