"""ctypes binding of libomnihd_hip.so — one prototype per symbol of include/omnihd_hip.h."""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_longlong, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# OMNIHD_LIB_PATH: load another build of the same library (experiments such as scripts/pool_traffic_abl.sh)
_LIB_PATH = os.environ.get("OMNIHD_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libomnihd_hip.so")

# name -> (restype, argtypes); the authoritative declarations are in include/omnihd_hip.h
PROTOTYPES = {
    "omnihd_version": (c_char_p, []),
    "omnihd_last_error": (c_char_p, []),
    "omnihd_device_count": (c_int, []),
    "omnihd_prefetch": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "omnihd_bev_pool_v2_fwd": (c_int, [c_void_p] * 8 + [c_int, c_int, c_void_p]),
    "omnihd_bev_pool_v2_bwd": (c_int, [c_void_p] * 10 + [c_int, c_int, c_void_p]),
    "omnihd_bev_pool_v2_fwd_csr": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p]),
    "omnihd_bev_pool_v2_fwd_direct": (c_int, [c_void_p] * 4 + [c_int, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "omnihd_bev_pool_v2_fwd_direct_dev": (c_int, [c_void_p] * 4 + [c_longlong, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]
                                          + [c_int] * 6 + [c_void_p]),
    "omnihd_pool_plan_sizes": (c_int, [c_longlong, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omnihd_pool_plan_build": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p] * 3 + [c_int, c_void_p, c_void_p, c_int, c_int]
                               + [c_void_p] * 11 + [c_size_t, c_void_p]),
    "omnihd_column_sums_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "omnihd_column_sums": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "omnihd_bev_pool_v2_bwd_patch": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_int64, c_void_p, c_void_p, c_int, c_void_p]),
    "omnihd_tile_desc": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "omnihd_csr_tiles_workspace_bytes": (c_size_t, [c_int]),
    "omnihd_csr_tiles": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "omnihd_bev_pool_v1_fwd": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "omnihd_bev_pool_v1_bwd": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "omnihd_bev_rank_keys": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_uint32, c_void_p]),
    "omnihd_sort_ranks_workspace_bytes": (c_size_t, [c_int64]),
    "omnihd_sort_ranks": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_uint32] + [c_void_p] * 9 +
                          [c_size_t, c_void_p]),
    "omnihd_ranks_feat_from_depth": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "omnihd_csr_from_sorted_keys": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "omnihd_permute_rows_zyx_to_yxz": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p,
                                               c_void_p]),
    "omnihd_voxelize_workspace_bytes": (c_size_t, [c_int]),
    "omnihd_voxelize_hard": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int] +
                             [c_void_p] * 6 + [c_size_t, c_void_p]),
    "omnihd_voxelize_grid_state_bytes": (c_size_t, [c_void_p, c_void_p]),
    "omnihd_voxelize_grid_state_init": (c_int, [c_void_p, c_size_t, c_void_p]),
    "omnihd_voxelize_grid_workspace_bytes": (c_size_t, [c_int]),
    "omnihd_voxelize_hard_grid": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 5 +
                                  [c_size_t, c_void_p, c_size_t, c_void_p]),
    "omnihd_pillar_scatter_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "omnihd_pillar_scatter": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_size_t, c_void_p]),
    "omnihd_pfn_channels": (c_int, [c_int, c_int]),
    "omnihd_pfn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "omnihd_pfn_moments": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_float] * 4 + [c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "omnihd_pfn_consts": (c_int, [c_void_p] * 4 + [c_int, c_longlong, c_float, c_float, c_int, c_int] + [c_void_p] * 4),
    "omnihd_pfn_apply": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_float] * 4 + [c_int] + [c_void_p] * 4),
    "omnihd_pfn_bwd_sums": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_float] * 4 + [c_int] + [c_void_p] * 5 + [c_size_t, c_void_p]),
    "omnihd_pfn_bwd_final": (c_int, [c_void_p] * 6 + [c_int, c_longlong, c_int] + [c_void_p] * 4),
    "omnihd_pillar_cell_map": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "omnihd_pillar_canvas": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "omnihd_conv_wgrad_nhwc_workspace_bytes": (c_size_t, [c_int] * 11),
    "omnihd_conv_wgrad_nhwc": (c_int, [c_void_p] * 5 + [c_int] * 11 + [c_void_p, c_size_t, c_void_p]),
    "omnihd_conv_wgrad_nhwc_f16": (c_int, [c_void_p] * 4 + [c_int] * 11 + [c_void_p, c_size_t, c_void_p]),
    "omnihd_cast_f16": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p]),
    "omnihd_conv_fwd_f16": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "omnihd_conv_fwd_supported": (c_int, [c_int] * 7),
    "omnihd_conv_fwd_bf16": (c_int, [c_void_p] * 4 + [c_int] * 8 + [c_void_p]),
    "omnihd_split_f32": (c_int, [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p]),
    "omnihd_conv_fwd_split": (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_void_p]),
    "omnihd_conv_dgrad_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "omnihd_anchor_loss_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "omnihd_anchor_loss_fwd": (c_int, [c_void_p] * 4 + [c_int] + [c_void_p] * 3 + [c_int] * 6 + [c_void_p, c_void_p, c_int, c_void_p, c_void_p]
                               + [c_void_p] * 5 + [c_size_t, c_void_p]),
    "omnihd_anchor_loss_bwd": (c_int, [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong] + [c_void_p] * 6),
    "omnihd_conv_gen_supported": (c_int, [c_int] * 12),
    "omnihd_conv_gen": (c_int, [c_int] + [c_void_p] * 6 + [c_int] * 11 + [c_void_p]),
    "omnihd_weight_images": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "omnihd_weight_images_cl": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "omnihd_dcn3x3_sample_fwd": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "omnihd_dcn3x3_sample_bwd": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "omnihd_dcn3x3_sample_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "omnihd_dcn3x3_sample_bwd_f32": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "omnihd_pillar_gather": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_void_p, c_void_p]),
    "omnihd_nms_rotated_workspace_bytes": (c_size_t, [c_int]),
    "omnihd_nms_rotated": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "omnihd_affine_act_fwd": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_void_p]),
    "omnihd_affine_act_bwd": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_void_p]),
    "omnihd_bn_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "omnihd_bn_channel_sums": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "omnihd_bn_fwd_consts": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_float, c_float, c_float, c_int] + [c_void_p] * 7),
    "omnihd_bn_bwd_consts": (c_int, [c_void_p] * 5 + [c_float, c_int] + [c_void_p] * 6),
    "omnihd_bn_bwd_apply": (c_int, [c_void_p] * 9 + [c_int64, c_int, c_void_p]),
    "omnihd_bn_train_fwd": (c_int, [c_void_p] * 6 + [c_float, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                    c_int, c_void_p, c_size_t, c_void_p]),
    "omnihd_bn_train_bwd": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 7 + [c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "omnihd_affine_act_fwd_f32": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_void_p]),
    "omnihd_affine_act_bwd_f32": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_void_p]),
    "omnihd_bn_channel_sums_f32": (c_int, [c_void_p] * 5 + [c_int64, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "omnihd_bn_bwd_apply_f32": (c_int, [c_void_p] * 9 + [c_int64, c_int, c_void_p]),
    "omnihd_bn_train_fwd_f32": (c_int, [c_void_p] * 6 + [c_float, c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                        c_int, c_void_p, c_size_t, c_void_p]),
    "omnihd_bn_train_bwd_f32": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 7 + [c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "omnihd_affine_act_fwd_f32_planes": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_int, c_void_p]),
    "omnihd_bn_train_fwd_f32_planes": (c_int, [c_void_p] * 6 + [c_float, c_float, c_float, c_int] + [c_void_p] * 5 + [c_int64, c_int,
                                               c_void_p, c_size_t, c_void_p]),
    "omnihd_bn_train_bwd_f32_planes": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 9 + [c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "omnihd_bn_train_bwd_f32_amax": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 8 + [c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "omnihd_affine_act_bwd_f32_amax": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_int, c_void_p]),
    "omnihd_radar_merge": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "omnihd_depth_head_fwd": (c_int, [c_void_p, c_longlong, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "omnihd_depth_head_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p, c_longlong, c_void_p, c_longlong, c_void_p]),
    "omnihd_iou_bev_matrix": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
}

_lib = None


def library_path():
    return _LIB_PATH


def lib():
    """Load (once) and return the library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with `make -C omnihd-scenes_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback for the HIP ops.")
        handle = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def require_gpu():
    """Fail loudly when the HIP path cannot run (no device)."""
    if lib().omnihd_device_count() <= 0:
        raise RuntimeError("omnihd_amd: no HIP device visible; the hot path has no CPU fallback")


def check(status, what):
    if status != 0:
        msg = lib().omnihd_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed with status {status}: {msg}")
