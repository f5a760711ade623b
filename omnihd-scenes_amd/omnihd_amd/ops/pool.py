"""bev_pool_v2 / bev_pool (v1) wrappers, the depth-head epilogue and rank preparation (csrc/bev_pool_v2.hip, bev_pool_v1.hip, rank_prep.hip, depth_head.hip).
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
from ._core import _on, _ptr, _same_device, _stream, _want, _workspace



# ---------------------------------------------------------------------------------------------
# bev_pool_v2
# ---------------------------------------------------------------------------------------------
def bev_pool_v2_forward(depth, feat, out, ranks_depth, ranks_feat, ranks_bev, interval_lengths,
                        interval_starts):
    """Same positional signature as the reference pybind function (lengths BEFORE starts):
    ops/bev_pool_v2/src/bev_pool.cpp:30-39.  Writes into ``out`` (pre-zeroed by the caller)."""
    _want(depth, torch.float32, "depth"); _want(feat, torch.float32, "feat")
    _want(out, torch.float32, "out")
    for n, t in (("ranks_depth", ranks_depth), ("ranks_feat", ranks_feat), ("ranks_bev", ranks_bev),
                 ("interval_lengths", interval_lengths), ("interval_starts", interval_starts)):
        _want(t, torch.int32, n)
    if feat.dim() != 5:
        raise ValueError("feat must be 5-D (B,N,H,W,C)")  # C = feat.size(4), bev_pool.cpp:40
    dev = _same_device(depth, feat, out, ranks_depth, ranks_feat, ranks_bev, interval_lengths, interval_starts)
    with _on(dev):
        check(lib().omnihd_bev_pool_v2_fwd(_ptr(depth), _ptr(feat), _ptr(ranks_depth), _ptr(ranks_feat),
                                           _ptr(ranks_bev), _ptr(interval_starts), _ptr(interval_lengths),
                                           _ptr(out), feat.size(4), interval_lengths.size(0), _stream()),
              "omnihd_bev_pool_v2_fwd")


def bev_pool_v2_backward(out_grad, depth_grad, feat_grad, depth, feat, ranks_depth, ranks_feat,
                         ranks_bev, interval_lengths, interval_starts):
    """Reference signature: ops/bev_pool_v2/src/bev_pool.cpp:74-85 (tables sorted by ranks_feat)."""
    for n, t in (("out_grad", out_grad), ("depth_grad", depth_grad), ("feat_grad", feat_grad),
                 ("depth", depth), ("feat", feat)):
        _want(t, torch.float32, n)
    for n, t in (("ranks_depth", ranks_depth), ("ranks_feat", ranks_feat), ("ranks_bev", ranks_bev),
                 ("interval_lengths", interval_lengths), ("interval_starts", interval_starts)):
        _want(t, torch.int32, n)
    if out_grad.dim() != 5:
        raise ValueError("out_grad must be 5-D (B,Z,Y,X,C)")  # C = out_grad.size(4), bev_pool.cpp:86
    dev = _same_device(out_grad, depth_grad, feat_grad, depth, feat, ranks_depth)
    with _on(dev):
        check(lib().omnihd_bev_pool_v2_bwd(_ptr(out_grad), _ptr(depth), _ptr(feat), _ptr(ranks_depth),
                                           _ptr(ranks_feat), _ptr(ranks_bev), _ptr(interval_starts),
                                           _ptr(interval_lengths), _ptr(depth_grad), _ptr(feat_grad),
                                           out_grad.size(4), interval_lengths.size(0), _stream()),
              "omnihd_bev_pool_v2_bwd")


def bev_pool_v2_forward_csr(depth, feat, ranks_depth, ranks_feat, row_ptr, out):
    """Dense forward for any channel count: every row of ``out`` (n_rows = row_ptr.numel()-1, C = feat.size(-1)) is written (see
    include/omnihd_hip.h: omnihd_bev_pool_v2_fwd_csr).  C = 64 runs :func:`bev_pool_v2_forward_direct` instead."""
    _want(depth, torch.float32, "depth"); _want(feat, torch.float32, "feat"); _want(out, torch.float32, "out")
    _want(ranks_depth, torch.int32, "ranks_depth"); _want(ranks_feat, torch.int32, "ranks_feat")
    _want(row_ptr, torch.int32, "row_ptr")
    c = feat.size(-1)
    n_rows = row_ptr.numel() - 1
    if out.numel() != n_rows * c:
        raise ValueError(f"out has {out.numel()} elements, expected {n_rows}*{c}")
    dev = _same_device(depth, feat, out, ranks_depth, ranks_feat, row_ptr)
    with _on(dev):
        check(lib().omnihd_bev_pool_v2_fwd_csr(_ptr(depth), _ptr(feat), _ptr(ranks_depth), _ptr(ranks_feat), _ptr(row_ptr), _ptr(out),
                                               c, n_rows, ranks_depth.numel(), _stream()), "omnihd_bev_pool_v2_fwd_csr")



_PREFETCH_STREAMS = {}


def prefetch(tensors):
    """Read-ahead of up to 4 static device tensors into the L2 / Infinity Cache on a side stream (a hint: see
    include/omnihd_hip.h, omnihd_prefetch).  Returns immediately; nothing waits for it."""
    ts = [t for t in tensors if t is not None and t.is_cuda and t.numel() > 0][:4]
    if not ts:
        return
    dev = ts[0].device
    side = _PREFETCH_STREAMS.get(dev.index)
    if side is None:
        side = _PREFETCH_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
    ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    sizes = (ctypes.c_size_t * len(ts))(*[t.numel() * t.element_size() for t in ts])
    # ordered behind the work already enqueued on the calling stream (the host runs milliseconds ahead of the device: an
    # unordered read-ahead would execute right away and be evicted again long before its consumer starts)
    side.wait_stream(torch.cuda.current_stream(dev))
    with _on(dev):
        check(lib().omnihd_prefetch(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(sizes, ctypes.c_void_p), len(ts),
                                    ctypes.c_void_p(side.cuda_stream)), "omnihd_prefetch")


def bev_pool_v2_forward_direct(depth, feat, pt, ivl_rel, desc32, row_ptr, out, depth_bins, feat_hw, empty_rows_kept=False):
    """Dense tiled forward for C = 64 whose lane groups walk their piece of the point list straight from global memory
    (see include/omnihd_hip.h: omnihd_bev_pool_v2_fwd_direct; tables from ``plan.direct_tables``)."""
    _want(depth, torch.float32, "depth"); _want(feat, torch.float32, "feat"); _want(out, torch.float32, "out")
    _want(pt, torch.int32, "pt"); _want(ivl_rel, torch.int32, "ivl_rel"); _want(desc32, torch.int32, "desc32")
    _want(row_ptr, torch.int32, "row_ptr")
    c = feat.size(-1)
    n_rows = row_ptr.numel() - 1
    if c != 64 or out.numel() != n_rows * c:
        raise ValueError(f"C must be 64 and out must hold {n_rows}*{c} elements")
    if desc32.dim() != 2 or desc32.size(1) != 32 or desc32.size(0) % 8:
        raise ValueError("desc32 must be (8*k, 32) int32")
    if depth.numel() != (feat.numel() // c) * int(depth_bins) or (feat.numel() // c) % int(feat_hw):
        raise ValueError(f"depth ({depth.numel()} values) must hold depth_bins={depth_bins} values per pixel row of feat "
                         f"({feat.numel() // c} rows, feat_hw={feat_hw})")
    dev = _same_device(depth, feat, out, pt, ivl_rel, desc32, row_ptr)
    with _on(dev):
        check(lib().omnihd_bev_pool_v2_fwd_direct(_ptr(depth), _ptr(feat), _ptr(pt), _ptr(ivl_rel), ivl_rel.numel(), _ptr(desc32),
                                                  desc32.size(0), _ptr(row_ptr), _ptr(out), c, n_rows, pt.numel(), int(depth_bins),
                                                  int(feat_hw), feat.numel() // c, 1 if empty_rows_kept else 0, _stream()),
              "omnihd_bev_pool_v2_fwd_direct")


def bev_pool_v2_backward_patch(out_grad, depth, feat, ranks_depth, ranks_row, pix_ptr, patch_order, depth_grad, feat_grad):
    """Patch backward for C = 64 (see include/omnihd_hip.h): writes BOTH gradients densely (no zero-fill by the caller).
    depth (B,N,D,H,W), feat (B,N,H,W,64), out_grad (n_rows, 64); tables sorted by pixel + their CSR ``pix_ptr``."""
    for n, t in (("out_grad", out_grad), ("depth", depth), ("feat", feat), ("depth_grad", depth_grad),
                 ("feat_grad", feat_grad)):
        _want(t, torch.float32, n)
    # ranks_depth None: ``ranks_row`` is the packed per-point table (output row | depth bin << 24), plan.bp_row_bin
    for n, t in (("ranks_row", ranks_row), ("pix_ptr", pix_ptr), ("patch_order", patch_order)) + ((("ranks_depth", ranks_depth),) if ranks_depth is not None else ()):
        _want(t, torch.int32, n)
    if depth.dim() != 5 or feat.dim() != 5 or feat.size(-1) != 64:
        raise ValueError("depth must be (B,N,D,H,W) and feat (B,N,H,W,64)")
    B, N, D, H, W = depth.shape
    n_img, fhw = B * N, H * W
    if pix_ptr.numel() != n_img * fhw + 1 or patch_order.numel() % 8:
        raise ValueError("pix_ptr must have B*N*H*W + 1 entries and patch_order 8*k")
    dev = _same_device(out_grad, depth, feat, depth_grad, feat_grad, ranks_row, pix_ptr, patch_order)
    if ranks_depth is None and D > 127:
        raise ValueError("the packed table holds at most 127 depth bins")
    with _on(dev):
        check(lib().omnihd_bev_pool_v2_bwd_patch(_ptr(out_grad), _ptr(depth), _ptr(feat), None if ranks_depth is None else _ptr(ranks_depth), _ptr(ranks_row),
                                                 _ptr(pix_ptr), _ptr(patch_order), patch_order.numel(), n_img, D, fhw,
                                                 out_grad.numel() // 64, _ptr(depth_grad), _ptr(feat_grad), 64, _stream()),
              "omnihd_bev_pool_v2_bwd_patch")


def tile_descriptors(row_ptr, tile_row, tile_order=None):
    """(8*ceil(n_tiles/8), 4) int32 launch schedule {first row, #rows, first point, #points}."""
    _want(row_ptr, torch.int32, "row_ptr"); _want(tile_row, torch.int32, "tile_row")
    n_tiles = tile_row.numel() - 1
    n_slots = 8 * ((n_tiles + 7) // 8)
    if tile_order is not None:
        _want(tile_order, torch.int32, "tile_order")
        if tile_order.numel() != n_slots:
            raise ValueError("tile_order must have 8*ceil(n_tiles/8) entries")
    desc = torch.empty((n_slots, 4), dtype=torch.int32, device=row_ptr.device)
    with _on(row_ptr.device):
        check(lib().omnihd_tile_desc(_ptr(row_ptr), _ptr(tile_row), _ptr(tile_order), n_tiles, _ptr(desc), _stream()),
              "omnihd_tile_desc")
    return desc


def csr_tiles(row_ptr, tile_items=768, long_len=512):
    """Tile table for the tiled dense forward (see include/omnihd_hip.h): int32 [n_tiles+1]."""
    _want(row_ptr, torch.int32, "row_ptr")
    n_rows = row_ptr.numel() - 1
    dev = row_ptr.device
    tile_row = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    h = ctypes.c_int(0)
    with _on(dev):
        ws = _workspace(lib().omnihd_csr_tiles_workspace_bytes(n_rows), dev)
        check(lib().omnihd_csr_tiles(_ptr(row_ptr), n_rows, tile_items, long_len, _ptr(tile_row), _ptr(count),
                                     ctypes.cast(ctypes.pointer(h), ctypes.c_void_p), _ptr(ws), ws.numel(), _stream()),
              "omnihd_csr_tiles")
    return tile_row[:h.value + 1].clone()


# ---------------------------------------------------------------------------------------------
# depth-head epilogue: softmax over D + depth / context split + the pooling's layouts (csrc/depth_head.hip)
# ---------------------------------------------------------------------------------------------
def _nhwc_rows(t, align_bytes):
    """(M, ch, H, W) tensor -> (tensor, row pitch in elements) such that pixel p's channels are the ``ch`` contiguous elements
    at p * pitch (channels-last memory, or a channel slice of a wider channels-last tensor); anything else is packed."""
    M, ch, H, W = t.shape
    P = t.stride(3) if W > 1 else (t.stride(2) if H > 1 else (t.stride(0) if M > 1 else ch))
    ok = ((ch == 1 or t.stride(1) == 1) and P >= ch and (W == 1 or t.stride(3) == P) and (H == 1 or t.stride(2) == W * P)
          and (M == 1 or t.stride(0) == H * W * P)
          and (t.data_ptr() % align_bytes == 0 and (P * t.element_size()) % align_bytes == 0))
    if ok:
        return t, P
    t = t.contiguous(memory_format=torch.channels_last)
    if t.data_ptr() % align_bytes or not t.is_contiguous(memory_format=torch.channels_last):
        t = t.clone(memory_format=torch.channels_last)
    return t, ch


class _DepthHead(torch.autograd.Function):
    """logits (M,D,H,W), context (M,C,H,W) or None, bf16 or fp32 ->
    depth (M,D,H,W) fp32 contiguous (= softmax(logits, 1)), depth_rows (M,H,W,D) fp32 or None, feat (M,H,W,C) fp32 or None."""

    @staticmethod
    def forward(ctx, logits, context, want_rows):
        M, D, H, W = logits.shape
        dev = logits.device
        is_f32 = logits.dtype == torch.float32
        lg, ld_l = _nhwc_rows(logits, 4 if is_f32 else 2)
        cx = ld_c = feat = None
        C = 0
        if context is not None:
            C = context.shape[1]
            cx, ld_c = _nhwc_rows(context, 16 if is_f32 else 8)
            feat = torch.empty((M, H, W, C), dtype=torch.float32, device=dev)
        depth = torch.empty((M, D, H, W), dtype=torch.float32, device=dev)
        rows = torch.empty((M, H, W, D), dtype=torch.float32, device=dev) if want_rows else None
        with _on(dev):
            check(lib().omnihd_depth_head_fwd(_ptr(lg), ld_l, _ptr(cx), ld_c or 0, 1 if is_f32 else 0, M, H * W, D, C,
                                              _ptr(depth), _ptr(rows), _ptr(feat), _stream()), "omnihd_depth_head_fwd")
        ctx.save_for_backward(depth)
        ctx.meta = (M, D, H, W, C, logits.dtype, context is not None)
        ctx.set_materialize_grads(False)
        return depth, rows, feat

    @staticmethod
    def backward(ctx, g_depth, g_rows, g_feat):
        (depth,) = ctx.saved_tensors
        M, D, H, W, C, dtype, has_ctx = ctx.meta
        dev = depth.device
        g_logits = g_ctx = None
        f32 = lambda t: None if t is None else t.contiguous().float()
        g_depth, g_rows, g_feat = f32(g_depth), f32(g_rows), f32(g_feat)
        want_ctx = has_ctx and ctx.needs_input_grad[1]
        if want_ctx:
            g_ctx = torch.empty((M, C, H, W), dtype=dtype, device=dev, memory_format=torch.channels_last)
            if g_feat is None:
                g_ctx.zero_()
        if ctx.needs_input_grad[0]:
            g_logits = torch.empty((M, D, H, W), dtype=dtype, device=dev, memory_format=torch.channels_last)
        if g_logits is not None or (want_ctx and g_feat is not None):
            scratch = g_logits if g_logits is not None else torch.empty((M, D, H, W), dtype=dtype, device=dev,
                                                                       memory_format=torch.channels_last)
            with _on(dev):
                check(lib().omnihd_depth_head_bwd(_ptr(depth), _ptr(g_depth), _ptr(g_rows), _ptr(g_feat),
                                                  1 if dtype == torch.float32 else 0, M, H * W, D, C, _ptr(scratch), D,
                                                  _ptr(g_ctx) if (want_ctx and g_feat is not None) else None, C, _stream()),
                      "omnihd_depth_head_bwd")
        return g_logits, g_ctx, None


def depth_head_supported(logits, context):
    return (logits.is_cuda and logits.dim() == 4 and logits.dtype in (torch.bfloat16, torch.float32) and logits.shape[1] <= 160
            and logits.shape[0] <= 65535 and context.dim() == 4 and context.dtype == logits.dtype and context.shape[1] % 4 == 0
            and context.shape[0] == logits.shape[0] and context.shape[2:] == logits.shape[2:])


def depth_head(logits, context, want_rows=False):
    """Depth-head epilogue of the LSS camera stream (reference cam_stream_lss_bevpoolv2_depthnet.py:134-143, :290):
    depth logits (M,D,H,W) + context (M,C,H,W), bf16 or fp32 ->
      depth      (M,D,H,W) fp32 contiguous, softmax over D             (view it (B,N,D,H,W): what bev_pool_v2 gathers from)
      depth_rows (M,H,W,D) fp32, the same values pixel-major, or None   (what the KL depth loss reads)
      feat       (M,H,W,C) fp32 contiguous context rows                 (view it (B,N,H,W,C): what bev_pool_v2 gathers from).
    A packed fp32 channels-last context tensor already IS ``feat``: it is returned as a view, no copy."""
    if not depth_head_supported(logits, context):
        raise TypeError("depth_head: (M,D,H,W) logits with D <= 160 and (M,C,H,W) context, C % 4 == 0, both bf16 or both fp32, "
                        "CUDA(HIP) tensors")
    M, C, H, W = context.shape
    if (context.dtype == torch.float32 and context.data_ptr() % 16 == 0 and C > 1 and H * W > 1
            and context.is_contiguous(memory_format=torch.channels_last)):
        depth, rows, _ = _DepthHead.apply(logits, None, bool(want_rows))
        return depth, rows, context.permute(0, 2, 3, 1)
    return _DepthHead.apply(logits, context, bool(want_rows))


# ---------------------------------------------------------------------------------------------
# bev_pool v1
# ---------------------------------------------------------------------------------------------
def bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, b, d, h, w):
    """Reference signature ops/bev_pool/src/bev_pool.cpp:22-28; allocates and returns [b,d,h,w,c]."""
    _want(x, torch.float32, "x"); _want(geom_feats, torch.int32, "geom_feats")
    _want(interval_lengths, torch.int32, "interval_lengths"); _want(interval_starts, torch.int32, "interval_starts")
    n, c = x.shape
    out = torch.zeros((b, d, h, w, c), dtype=x.dtype, device=x.device)
    with _on(x.device):
        check(lib().omnihd_bev_pool_v1_fwd(_ptr(x), _ptr(geom_feats), _ptr(interval_starts),
                                           _ptr(interval_lengths), _ptr(out), b, d, h, w, n, c,
                                           interval_lengths.size(0), _stream()), "omnihd_bev_pool_v1_fwd")
    return out


def bev_pool_backward(out_grad, geom_feats, interval_lengths, interval_starts, b, d, h, w):
    """Reference signature ops/bev_pool/src/bev_pool.cpp:60-66; returns x_grad [n,c]."""
    _want(out_grad, torch.float32, "out_grad"); _want(geom_feats, torch.int32, "geom_feats")
    _want(interval_lengths, torch.int32, "interval_lengths"); _want(interval_starts, torch.int32, "interval_starts")
    n = geom_feats.size(0)
    c = out_grad.size(4)
    x_grad = torch.zeros((n, c), dtype=out_grad.dtype, device=out_grad.device)
    with _on(out_grad.device):
        check(lib().omnihd_bev_pool_v1_bwd(_ptr(out_grad), _ptr(geom_feats), _ptr(interval_starts),
                                           _ptr(interval_lengths), _ptr(x_grad), b, d, h, w, n, c,
                                           interval_lengths.size(0), _stream()), "omnihd_bev_pool_v1_bwd")
    return x_grad


# ---------------------------------------------------------------------------------------------
# rank tables
# ---------------------------------------------------------------------------------------------
def _bits_for(max_value):
    return max(1, int(max_value).bit_length())


def sort_ranks(keys, payloads, key_bits, sentinel=0xFFFFFFFF):
    """Stable sort by key + run-length encode.  ``keys``: int32/uint32-as-int32 tensor; payloads:
    up to three int32 tensors.  Returns (keys_sorted, payloads_sorted, starts, lengths) trimmed to
    the non-sentinel part (one host sync to read the two counts)."""
    _want(keys, torch.int32, "keys")
    n = keys.numel()
    dev = keys.device
    pl = list(payloads) + [None] * (3 - len(payloads))
    for i, p in enumerate(pl):
        if p is not None:
            _want(p, torch.int32, f"payload{i}")
    keys_out = torch.empty_like(keys)
    outs = [torch.empty_like(p) if p is not None else None for p in pl]
    starts = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    lengths = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    h_counts = (ctypes.c_int * 2)(0, 0)
    with _on(dev):
        ws_bytes = lib().omnihd_sort_ranks_workspace_bytes(n)
        if ws_bytes == 0:
            check(-4, "omnihd_sort_ranks_workspace_bytes")
        ws = _workspace(ws_bytes, dev)
        check(lib().omnihd_sort_ranks(_ptr(keys), _ptr(pl[0]), _ptr(pl[1]), _ptr(pl[2]), n, key_bits,
                                      sentinel, _ptr(keys_out), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]),
                                      _ptr(starts), _ptr(lengths), _ptr(counts),
                                      ctypes.cast(h_counts, ctypes.c_void_p), _ptr(ws), ws.numel(), _stream()),
              "omnihd_sort_ranks")
    n_pts, n_int = int(h_counts[0]), int(h_counts[1])
    return (keys_out[:n_pts], [o[:n_pts] if o is not None else None for o in outs[:len(payloads)]],
            starts[:n_int], lengths[:n_int])


def rank_keys(geom, dx, bx, nx):
    """Frustum geometry (B,N,D,H,W,3) fp32 -> (keys int32 [Ntot], idx int32 [Ntot], sentinel).
    ``dx``/``bx`` fp32 and ``nx`` integer triples as produced by the reference's gen_dx_bx
    (cam_stream_lss_bevpoolv2_depthnet.py:80-85)."""
    _want(geom, torch.float32, "geom")
    if geom.dim() != 6 or geom.size(-1) != 3:
        raise ValueError("geom must be (B,N,D,H,W,3)")
    B = geom.size(0)
    n_total = geom.numel() // 3
    dx = np.asarray(dx, dtype=np.float32)
    bx = np.asarray(bx, dtype=np.float32)
    nx = np.asarray(nx, dtype=np.int64)
    off = (bx - dx / np.float32(2.0)).astype(np.float32)      # the reference's fp32 tensor arithmetic
    n_vox = int(B * nx[0] * nx[1] * nx[2])
    if n_vox >= 2 ** 31 - 1 or n_total >= 2 ** 31:
        raise ValueError("grid too large for int32 rank tables")
    keys = torch.empty(n_total, dtype=torch.int32, device=geom.device)
    idx = torch.empty(n_total, dtype=torch.int32, device=geom.device)
    h_off = (ctypes.c_float * 3)(*off.tolist())
    h_dx = (ctypes.c_float * 3)(*dx.tolist())
    h_nx = (ctypes.c_int * 3)(*[int(v) for v in nx])
    with _on(geom.device):
        check(lib().omnihd_bev_rank_keys(_ptr(geom), n_total, n_total // B,
                                         ctypes.cast(h_off, ctypes.c_void_p), ctypes.cast(h_dx, ctypes.c_void_p),
                                         ctypes.cast(h_nx, ctypes.c_void_p), _ptr(keys), _ptr(idx), n_vox, _stream()),
              "omnihd_bev_rank_keys")
    return keys, idx, n_vox


def ranks_feat_from_depth(ranks_depth, d, hw):
    _want(ranks_depth, torch.int32, "ranks_depth")
    out = torch.empty_like(ranks_depth)
    with _on(ranks_depth.device):
        check(lib().omnihd_ranks_feat_from_depth(_ptr(ranks_depth), ranks_depth.numel(), d, hw, _ptr(out), _stream()),
              "omnihd_ranks_feat_from_depth")
    return out


def csr_from_sorted_keys(sorted_keys, n_rows):
    _want(sorted_keys, torch.int32, "sorted_keys")
    row_ptr = torch.empty(n_rows + 1, dtype=torch.int32, device=sorted_keys.device)
    with _on(sorted_keys.device):
        check(lib().omnihd_csr_from_sorted_keys(_ptr(sorted_keys), sorted_keys.numel(), n_rows, _ptr(row_ptr), _stream()),
              "omnihd_csr_from_sorted_keys")
    return row_ptr


def permute_rows_zyx_to_yxz(rows, nz, ny, nx):
    _want(rows, torch.int32, "rows")
    out = torch.empty_like(rows)
    with _on(rows.device):
        check(lib().omnihd_permute_rows_zyx_to_yxz(_ptr(rows), rows.numel(), nz, ny, nx, _ptr(out), _stream()),
              "omnihd_permute_rows_zyx_to_yxz")
    return out


def voxel_pooling_prepare_v2(coor, dx, bx, nx):
    """Device implementation of the reference's ``voxel_pooling_prepare_v2``
    (cam_stream_lss_bevpoolv2_depthnet.py:302-362): five int32 tables in canonical (stable) order,
    or five ``None`` when no frustum point falls inside the grid."""
    B, N, D, H, W, _ = coor.shape
    keys, idx, sentinel = rank_keys(coor, dx, bx, nx)
    ranks_bev, (ranks_depth,), starts, lengths = sort_ranks(keys, [idx], _bits_for(sentinel), sentinel)
    if ranks_bev.numel() == 0:
        return None, None, None, None, None
    ranks_feat = ranks_feat_from_depth(ranks_depth.contiguous(), D, H * W)
    return (ranks_bev.contiguous(), ranks_depth.contiguous(), ranks_feat, starts.contiguous(),
            lengths.contiguous())


def backward_tables(ranks_bev, ranks_depth, ranks_feat, n_feat_rows=None):
    """The re-sort of QuickCumsumCuda.backward (ops/bev_pool_v2/bev_pool.py:47-57) on the device:
    stable sort by ranks_feat; returns (ranks_bev, ranks_depth, ranks_feat, starts, lengths)."""
    if n_feat_rows is None:
        n_feat_rows = int(ranks_feat.max().item()) + 1 if ranks_feat.numel() else 1
    rf, (rd, rb), starts, lengths = sort_ranks(ranks_feat.contiguous(), [ranks_depth.contiguous(), ranks_bev.contiguous()],
                                               _bits_for(n_feat_rows))
    return rb.contiguous(), rd.contiguous(), rf.contiguous(), starts.contiguous(), lengths.contiguous()
