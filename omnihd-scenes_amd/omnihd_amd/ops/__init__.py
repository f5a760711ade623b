"""Tensor-level wrappers over the C ABI (argument checks + pointer / stream hand-over only), by concern:
  _core         Shared plumbing of the operator wrappers: raw stream handle, device guard, argument checks, workspaces, live counters.
  pool          bev_pool_v2 / bev_pool (v1) wrappers, the depth-head epilogue and rank preparation (csrc/bev_pool_v2.hip, bev_pool_v1.hip, rank_prep.hip, depth_head.hip).
  radar         Radar branch: hard voxelisation, pillar scatter, fused pillar feature net, sweep merge (csrc/voxelize.hip, pillar_scatter.hip, pillar_pfn.hip, radar_merge.hip).
  conv_kernels  Convolution kernels behind the C ABI: weight gradients, implicit-GEMM forward / data gradient (bf16, split, half), the general strided kernel, column sums.
  policy        Which implementation runs a convolution: the persisted / measured choice tables and the environment policies (OMNIHD_CONV_POLICY, OMNIHD_FP32_CONV).
  planes        Operand planes of fp32 activations: the hi / lo bf16 split, the IEEE-half cast with its device-side scale, and the producer -> consumer hand-over tags.
  weights       Weight images (bf16, split, half; forward and data-gradient layouts) kept current behind the optimiser step with one launch.
  streams       Weight gradients on a side stream (plain and under DistributedDataParallel bucket views) and the live fast-path report.
  conv_fp32     The fp32 step's convolutions as autograd Functions: fp32-grade split form (_ConvSplit) and TF32-grade half form (_ConvF16).
  conv_bf16     The bf16 step's convolutions and transposed convolutions as autograd Functions, bias gradients by column sums.
  norm          BatchNorm epilogues as autograd Functions: frozen (affine + residual + ReLU) and training mode (csrc/affine_act.hip, batch_norm.hip).
  misc          Deformable-convolution sampling, rotated NMS, fused anchor targets + detection losses (csrc/dcn_sample.hip, nms_rotated.hip, anchor_loss.hip).
Every top-level name of every part is re-exported here, private ones included: `ops.X` is the public face (tests spy on and
replace functions through it); state that code REBINDS lives in one part and is reached through that part."""
from . import _core
from ._core import (_ptr, _current_device, _raw_stream, _stream, _want, _same_device, _workspace, _NULL_CTX, _SIZE_CACHE, _on,
    _WGRAD_WS, _wgrad_workspace, _want_cl, _pair_same, deterministic, FAST_PATHS, _CL, _rows_view, _f32c)
from . import pool
from .pool import (bev_pool_v2_forward, bev_pool_v2_backward, bev_pool_v2_forward_csr, _PREFETCH_STREAMS, prefetch,
    bev_pool_v2_forward_direct, bev_pool_v2_backward_patch, tile_descriptors, csr_tiles, _nhwc_rows, _DepthHead,
    depth_head_supported, depth_head, bev_pool_forward, bev_pool_backward, _bits_for, sort_ranks, rank_keys,
    ranks_feat_from_depth, csr_from_sorted_keys, permute_rows_zyx_to_yxz, voxel_pooling_prepare_v2, backward_tables)
from . import radar
from .radar import (PendingVoxels, _VOXEL_STATE, hard_voxelize_async, hard_voxelize, _SCATTER_MAPS, _PillarScatter,
    pillar_scatter, PFN_CLUSTER, PFN_CENTER, PFN_DISTANCE, PFN_LEGACY, PFN_RADAR, pfn_channels, _FusedPFN, pfn_fused,
    radar_merge)
from . import conv_kernels
from .conv_kernels import (conv_wgrad, conv_wgrad_split, wgrad_nhwc_preferred, _NHWC_OK, conv3x3_wgrad, conv1x1_wgrad,
    conv_wgrad_supported, conv3x3_wgrad_supported, conv_fwd_supported, CONV_TIMING, CONV_TIMING_GEOMETRY, _conv_timed,
    conv_fwd, conv_dgrad_weights, _GEN_OK, _conv_out_hw, conv_gen_supported, conv_gen, conv_fwd_split, conv_split_geometry,
    conv_fwd_f16, conv_wgrad_f16, column_sums)
from . import policy
from .policy import (_ChoiceTable, _CHOICE_INFO, _PERSISTED, _tuplify, _persisted_choices, _device_arch, choice_table_info,
    save_choice_table, _WGRAD_CHOICE, _miopen_wgrad, _tuned_wgrad, wgrad_choice_for, conv_all_miopen, wgrad_choices,
    _CONV_CHOICE, _CONV_IMPLS, _conv_policy, _conv_impl, conv_choices, sync_tuned_choices, f16_handover, _SPLIT_CHOICE, _clock,
    _fp32_policy, _split_pick, split_choices, conv_split_all_miopen)
from . import planes
from .planes import (split_f32, _alloc_planes, _PLANES_WANTED, _PLANES_UNUSED, HANDOVER_STATS, planes_wanted, tag_planes,
    tag_producer, _HALF_WANTED, half_wanted, tag_half, take_half, take_planes, cast_f16, _AMAX_RING, _AMAX_SLOTS, _amax_slot,
    _f16_plane)
from . import weights
from .weights import (_WEIGHT_GEN, _bump_weight_generation, _WEIGHT_GEN_HOOK, weights_changed, _wver, _BF16_SHADOW,
    _SHADOW_EPOCH, bf16_of, _BF16_DGRAD, bf16_dgrad_image, _Bf16Weight, bf16_weight, _BF16_PLAN, refresh_bf16_shadows,
    _SPLIT_SHADOW, split_dgrad_weights, split_weight, _WIMG_DTYPE, _WIMG_TABLES, WIMG_STATS, _weight_image_table,
    weight_images, refresh_split_shadows, _F16_SHADOW, f16_weight, refresh_f16_shadows)
from . import streams
from .streams import (fast_paths_report, fast_paths_reset, _WGRAD_SIDE, _WGRAD_SIDE_USED, _WGRAD_SEEN, _VIEW_WRITTEN,
    _WGRAD_ARMED, _WGRAD_PASS, _WGRAD_ENGINE_OK, _DDP, ddp_wgrad_overlap, _dense, _ddp_bucket_hook, _ddp_bucket_view,
    _view_writable, ddp_overlap_info, _wgrad_side_stream, _wgrad_pass_begin, wgrad_overlap_join, wgrad_overlap_arm,
    wgrad_overlap_fence)
from . import conv_fp32
from .conv_fp32 import (_ConvSplit, conv_split_supported, conv_split, conv_f16_applies, _ConvF16, conv_grad_planes_ok)
from . import conv_bf16
from .conv_bf16 import (_ConvBiasColsum, wgrad_split_padded, conv_bias_colsum_supported, conv_bias_colsum, _ConvHipWgrad,
    conv3x3, conv_hip_wgrad, _gen_or_library, _deconv_as_conv, _DeconvHipWgrad, _DeconvSplit, deconv_split_supported,
    deconv_split, deconv_supported, deconv_hip_wgrad)
from . import norm
from .norm import (_AffineAct, affine_act_supported, affine_act, _BnTrainAct, bn_train_supported, bn_train_act)
from . import misc
from .misc import (_DcnSample, dcn3x3_sample, dcn3x3_supported, nms_rotated, iou_bev_matrix, _AnchorLoss, anchor_loss)

PARTS = (_core, pool, radar, conv_kernels, policy, planes, weights, streams, conv_fp32, conv_bf16, norm, misc)
