"""Tensor-level wrappers over the C ABI (argument checks + pointer/stream hand-over only)."""
import contextlib
import ctypes
import os
from ._env import env as _env
import weakref

import numpy as np
import torch

from ._lib import check, lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# torch.cuda.current_stream() builds a Python Stream object through three layers of argument checking (~9 us; the step
# asks ~380 times); the raw handle of the current stream of the current device is one C call.
_current_device = torch._C._cuda_getDevice if hasattr(torch._C, "_cuda_getDevice") else torch.cuda.current_device
if hasattr(torch._C, "_cuda_getCurrentRawStream"):
    def _raw_stream():
        return torch._C._cuda_getCurrentRawStream(_current_device())
else:                                                           # pragma: no cover
    def _raw_stream():
        return torch.cuda.current_stream().cuda_stream


def _stream():
    return ctypes.c_void_p(_raw_stream())


def _want(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA(HIP) tensor; the HIP ops have no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _same_device(*ts):
    dev = ts[0].device
    for t in ts:
        if t is not None and t.device != dev:
            raise ValueError("all tensors must be on the same device")
    return dev


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# The training step issues a few hundred of these calls; at ~45 ms per step the Python side must stay cheap:
# no device context switch when the tensor's device is already current, size queries cached per geometry.

_NULL_CTX = contextlib.nullcontext()
_SIZE_CACHE = {}


def _on(dev):
    return _NULL_CTX if _current_device() == dev.index else torch.cuda.device(dev)


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


# ---------------------------------------------------------------------------------------------
# radar: hard voxelisation + pillar scatter
# ---------------------------------------------------------------------------------------------
class PendingVoxels:
    """Voxelisation that has been enqueued; ``get()`` waits only for ITS event (the voxel count travelling to pinned
    host memory), not for the device queue — so work enqueued in between (the image branch) is not drained."""

    def __init__(self, voxels, coors, num_points, host_count, event):
        self.voxels, self.coors, self.num_points, self.host_count, self.event = voxels, coors, num_points, host_count, event

    def get(self):
        self.event.synchronize()
        m = int(self.host_count[0])
        return self.voxels[:m], self.coors[:m], self.num_points[:m]


_VOXEL_STATE = {}


def hard_voxelize_async(points, voxel_size, point_cloud_range, max_points, max_voxels):
    """One sample: points (N,F) fp32 -> PendingVoxels of (voxels (M,max_points,F), coors (M,3)=(z,y,x) int32,
    num_points (M,) int32).  mmdet3d Voxelization semantics (see include/omnihd_hip.h)."""
    _want(points, torch.float32, "points")
    n, f = points.shape
    dev = points.device
    voxels = torch.empty((max_voxels, max_points, f), dtype=torch.float32, device=dev)
    coors = torch.empty((max_voxels, 3), dtype=torch.int32, device=dev)
    num_points = torch.empty((max_voxels,), dtype=torch.int32, device=dev)
    voxel_num = torch.empty(1, dtype=torch.int32, device=dev)          # written by both paths (also for n = 0)
    h_vs = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in voxel_size])
    h_rg = (ctypes.c_float * 6)(*[float(np.float32(v)) for v in point_cloud_range])
    with _on(dev):
        st = _raw_stream()
        L = lib()
        state = None
        if max_points <= 16 and _env("OMNIHD_VOXELIZE_GRID", "1") != "0":
            # three launches on a persistent per-cell state (idle between calls), for grids of up to 4 M cells
            skey = (dev.index, st, tuple(h_vs), tuple(h_rg))
            ent = _VOXEL_STATE.get(skey)
            if ent is None or ent[1]:
                nbytes = L.omnihd_voxelize_grid_state_bytes(ctypes.cast(h_vs, ctypes.c_void_p), ctypes.cast(h_rg, ctypes.c_void_p))
                if nbytes:
                    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                    check(L.omnihd_voxelize_grid_state_init(_ptr(buf), nbytes, st), "omnihd_voxelize_grid_state_init")
                    ent = _VOXEL_STATE[skey] = [buf, False]
                else:
                    ent = _VOXEL_STATE[skey] = [None, False]
            state = ent[0]
        if state is not None:
            ws = _workspace(L.omnihd_voxelize_grid_workspace_bytes(n), dev)
            ent[1] = True                          # a call that fails in between leaves the state dirty: rebuilt next time
            check(L.omnihd_voxelize_hard_grid(_ptr(points), n, f, ctypes.cast(h_vs, ctypes.c_void_p), ctypes.cast(h_rg, ctypes.c_void_p),
                                              max_points, max_voxels, _ptr(voxels), _ptr(coors), _ptr(num_points), _ptr(voxel_num),
                                              _ptr(state), state.numel(), _ptr(ws), ws.numel(), st), "omnihd_voxelize_hard_grid")
            ent[1] = False
        else:
            ws_bytes = L.omnihd_voxelize_workspace_bytes(n)
            if ws_bytes == 0:
                check(-4, "omnihd_voxelize_workspace_bytes")
            ws = _workspace(ws_bytes, dev)
            check(L.omnihd_voxelize_hard(_ptr(points), n, f, ctypes.cast(h_vs, ctypes.c_void_p),
                                         ctypes.cast(h_rg, ctypes.c_void_p), max_points, max_voxels,
                                         _ptr(voxels), _ptr(coors), _ptr(num_points), _ptr(voxel_num), None,
                                         _ptr(ws), ws.numel(), st), "omnihd_voxelize_hard")
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(voxel_num, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    return PendingVoxels(voxels, coors, num_points, host, ev)


def hard_voxelize(points, voxel_size, point_cloud_range, max_points, max_voxels):
    """Synchronous form of ``hard_voxelize_async``."""
    return hard_voxelize_async(points, voxel_size, point_cloud_range, max_points, max_voxels).get()


_SCATTER_MAPS = {}


class _PillarScatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, coors, batch, ny, nx, channels_last):
        feats = feats.contiguous().float()
        coors = coors.contiguous().int()
        m, c = feats.shape
        dev = feats.device
        shape = (batch, ny, nx, c) if channels_last else (batch, c, ny, nx)
        canvas = torch.empty(shape, dtype=torch.float32, device=dev)
        with _on(dev):
            # a cell map per (device, stream, grid) that is all -1 between calls: the canvas kernel resets what the map kernel
            # entered, so the steady state is two launches (no memset); `dirty` covers a call that failed in between
            st = _raw_stream()
            key = (dev.index, st, batch, ny, nx)
            ent = _SCATTER_MAPS.get(key)
            if ent is None or ent[1]:
                ent = _SCATTER_MAPS[key] = [torch.full((batch * ny * nx,), -1, dtype=torch.int32, device=dev), False]
            ent[1] = True
            check(lib().omnihd_pillar_cell_map(_ptr(coors), m, batch, ny, nx, _ptr(ent[0]), st), "omnihd_pillar_cell_map")
            check(lib().omnihd_pillar_canvas(_ptr(feats), _ptr(ent[0]), c, batch, ny, nx, 1 if channels_last else 0, 1,
                                             _ptr(canvas), st), "omnihd_pillar_canvas")
            ent[1] = False
        ctx.save_for_backward(coors)
        ctx.meta = (m, c, batch, ny, nx, channels_last)
        if channels_last:
            canvas = canvas.permute(0, 3, 1, 2)   # logical NCHW view over NHWC memory
        return canvas

    @staticmethod
    def backward(ctx, g):
        (coors,) = ctx.saved_tensors
        m, c, batch, ny, nx, channels_last = ctx.meta
        # the gather kernel reads either memory layout of the (B,C,ny,nx) gradient: take it as it comes
        g = g.float()
        nhwc = g.is_contiguous(memory_format=torch.channels_last) and not g.is_contiguous()
        if not nhwc:
            g = g.contiguous()
        fg = torch.empty((m, c), dtype=torch.float32, device=g.device)
        with _on(g.device):
            check(lib().omnihd_pillar_gather(_ptr(g), _ptr(coors), m, c, batch, ny, nx,
                                             1 if nhwc else 0, _ptr(fg), _stream()),
                  "omnihd_pillar_gather")
        return fg, None, None, None, None, None


def pillar_scatter(feats, coors, batch, ny, nx, channels_last=False):
    """(M,C) pillar features + (M,4)=(b,z,y,x) coords -> dense (B,C,ny,nx) canvas (differentiable
    w.r.t. feats).  With ``channels_last`` the memory layout is NHWC under an NCHW-shaped view."""
    if not feats.is_cuda:
        raise RuntimeError("pillar_scatter: CUDA(HIP) tensors only; no CPU path")
    return _PillarScatter.apply(feats.float(), coors, int(batch), int(ny), int(nx), bool(channels_last))


# ---------------------------------------------------------------------------------------------
# fused pillar feature net (csrc/pillar_pfn.hip)
# ---------------------------------------------------------------------------------------------
PFN_CLUSTER, PFN_CENTER, PFN_DISTANCE, PFN_LEGACY, PFN_RADAR = 1, 2, 4, 8, 16


def pfn_channels(raw_channels, flags):
    """Decorated channels K of a pillar point for ``flags`` (what the fused kernels support: K <= 16)."""
    return int(lib().omnihd_pfn_channels(int(raw_channels), int(flags)))


class _FusedPFN(torch.autograd.Function):
    """out (M, 64) = max over the slots of a pillar of relu(BatchNorm(W x)), x = the decorated point (see csrc/pillar_pfn.hip).
    Training statistics come from the moments of x; with a process group they are averaged over the ranks (mean of rank means,
    the reference's naiveSyncBN semantics, ops/norm.py:65-72) by ONE all-reduce of K + K*K doubles forward and one of 128 floats
    backward.  Gradients flow to weight / gamma / beta (the points carry none in the reference either)."""

    @staticmethod
    def forward(ctx, voxels, num_points, coors, weight, gamma, beta, running_mean, running_var, geom, flags, eps, momentum,
                training, group):
        voxels = voxels.contiguous().float()
        num_points = num_points.contiguous().int()
        coors = coors.contiguous().int()
        w = weight.detach().contiguous().float()
        ga, be = gamma.detach().contiguous().float(), beta.detach().contiguous().float()
        m, p, f = voxels.shape
        vx, vy, x_off, y_off = (float(v) for v in geom)
        k = w.shape[1]
        dev = voxels.device
        L = lib()
        out = torch.empty((m, 64), dtype=torch.float32, device=dev)
        consts = torch.empty(256, dtype=torch.float32, device=dev)
        world = 1
        if training and group is not None and torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size(group)
        moments = None
        with _on(dev):
            st = _raw_stream()
            if training:
                if m == 0:
                    raise RuntimeError("fused pillar feature net: BatchNorm in training mode needs at least one pillar")
                moments = torch.empty(k + k * k, dtype=torch.float64, device=dev)
                ws = _workspace(L.omnihd_pfn_workspace_bytes(m, p, k), dev)
                check(L.omnihd_pfn_moments(_ptr(voxels), _ptr(num_points), _ptr(coors), m, p, f, vx, vy, x_off, y_off, flags,
                                           _ptr(moments), _ptr(ws), ws.numel(), st), "omnihd_pfn_moments")
                stats = moments
                if world > 1:                      # mean of the ranks' means / mean squares: every rank weighs 1 / world
                    stats = moments.clone()
                    torch.distributed.all_reduce(stats, group=group)
                    stats.mul_(1.0 / world)
                check(L.omnihd_pfn_consts(_ptr(w), _ptr(ga), _ptr(be), _ptr(stats), k, m * p, float(eps), float(momentum),
                                          1 if world == 1 else 0, 0, _ptr(running_mean), _ptr(running_var), _ptr(consts), st),
                      "omnihd_pfn_consts")
            else:
                check(L.omnihd_pfn_consts(_ptr(w), _ptr(ga), _ptr(be), None, k, max(m * p, 1), float(eps), 0.0, 0, 1,
                                          _ptr(running_mean), _ptr(running_var), _ptr(consts), st), "omnihd_pfn_consts")
            if m:
                check(L.omnihd_pfn_apply(_ptr(voxels), _ptr(num_points), _ptr(coors), m, p, f, vx, vy, x_off, y_off, flags,
                                         _ptr(w), _ptr(consts), _ptr(out), st), "omnihd_pfn_apply")
        if moments is None:                        # inference statistics: the backward's batch terms vanish (world = 0 says so)
            moments = torch.zeros(k + k * k, dtype=torch.float64, device=dev)
        ctx.save_for_backward(voxels, num_points, coors, w, ga, consts, moments)
        ctx.meta = (vx, vy, x_off, y_off, flags, world if training else 0, group)
        return out

    @staticmethod
    def backward(ctx, g):
        voxels, num_points, coors, w, ga, consts, moments = ctx.saved_tensors
        vx, vy, x_off, y_off, flags, world, group = ctx.meta
        m, p, f = voxels.shape
        k = w.shape[1]
        dev = voxels.device
        g = g.contiguous().float()
        L = lib()
        if m == 0:
            return (None, None, None, torch.zeros_like(w), torch.zeros_like(ga), torch.zeros_like(ga)) + (None,) * 8
        sums = torch.empty(128 + 64 * k, dtype=torch.float32, device=dev)
        dw = torch.empty((64, k), dtype=torch.float32, device=dev)
        dg, db = torch.empty(64, dtype=torch.float32, device=dev), torch.empty(64, dtype=torch.float32, device=dev)
        with _on(dev):
            st = _raw_stream()
            ws = _workspace(L.omnihd_pfn_workspace_bytes(m, p, k), dev)
            check(L.omnihd_pfn_bwd_sums(_ptr(voxels), _ptr(num_points), _ptr(coors), m, p, f, vx, vy, x_off, y_off, flags, _ptr(w),
                                        _ptr(consts), _ptr(g), _ptr(sums), _ptr(ws), ws.numel(), st), "omnihd_pfn_bwd_sums")
            ab = sums
            if world > 1:
                ab = sums[:128].clone()
                torch.distributed.all_reduce(ab, group=group)
            check(L.omnihd_pfn_bwd_final(_ptr(sums), _ptr(ab), _ptr(moments), _ptr(w), _ptr(ga), _ptr(consts), k, m * p, world,
                                         _ptr(dw), _ptr(dg), _ptr(db), st), "omnihd_pfn_bwd_final")
        return (None, None, None, dw, dg, db) + (None,) * 8


def pfn_fused(voxels, num_points, coors, weight, norm, geom, flags, group=None):
    """Fused PillarFeatureNet layer: ``weight`` (64, K) fp32, ``norm`` a BatchNorm-type module with 64 channels (its affine
    parameters, running statistics, eps, momentum and train / eval state are used), ``geom`` = (vx, vy, x_offset, y_offset)."""
    if not voxels.is_cuda:
        raise RuntimeError("pfn_fused: CUDA(HIP) tensors only; no CPU path")
    training = bool(norm.training or not norm.track_running_stats)
    if training and norm.track_running_stats and norm.num_batches_tracked is not None:
        norm.num_batches_tracked.add_(1)
    momentum = 0.0 if norm.momentum is None else norm.momentum
    return _FusedPFN.apply(voxels, num_points, coors, weight, norm.weight, norm.bias, norm.running_mean, norm.running_var,
                           tuple(geom), int(flags), norm.eps, momentum, training, group)


# ---------------------------------------------------------------------------------------------
# dense BEV convolutions: hand-written MFMA weight gradient
# ---------------------------------------------------------------------------------------------
_WGRAD_WS = {}


def _wgrad_workspace(nbytes, dev):
    key = (dev.index, _raw_stream())
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _workspace(nbytes, dev)
        _WGRAD_WS[key] = ws
    return ws


def _want_cl(t, name):
    if not t.is_cuda or t.dtype != torch.bfloat16 or t.dim() != 4:
        raise TypeError(f"{name} must be a 4-D bf16 CUDA(HIP) tensor")
    if not t.is_contiguous(memory_format=torch.channels_last):
        raise ValueError(f"{name} must be channels-last contiguous")


def conv_wgrad(x, grad_out, kernel_size, stride=1, padding=0, dilation=1):
    """Weight gradient of a dense Conv2d on the matrix cores: x (B,Cin,H,W) and grad_out (B,Cout,Ho,Wo) bf16
    channels-last -> dW (Cout,Cin,k,k) fp32 in channels-last memory.  k in {1,3}, square stride/dilation."""
    _want_cl(x, "x"); _want_cl(grad_out, "grad_out")
    B, cin, H, W = x.shape
    _, cout, Ho, Wo = grad_out.shape
    k = int(kernel_size)
    dev = x.device
    dw = torch.empty((cout, k, k, cin), dtype=torch.float32, device=dev)
    geo = (B, H, W, cin, Ho, Wo, cout, k, k, int(stride), int(padding), int(dilation))
    L = lib()
    if wgrad_nhwc_preferred(B, H, W, cin, Ho, Wo, cout, k, int(stride), int(padding), int(dilation)):
        g11 = (B, H, W, cin, Ho, Wo, cout, k, int(stride), int(padding), int(dilation))
        with _on(dev):
            nbytes = _SIZE_CACHE.get(("nhwc",) + g11)
            if nbytes is None:
                nbytes = _SIZE_CACHE[("nhwc",) + g11] = L.omnihd_conv_wgrad_nhwc_workspace_bytes(*g11)
            ws = _wgrad_workspace(nbytes, dev)
            check(L.omnihd_conv_wgrad_nhwc(x.data_ptr(), None, grad_out.data_ptr(), None, dw.data_ptr(), *g11, ws.data_ptr(), ws.numel(),
                                           _raw_stream()), "omnihd_conv_wgrad_nhwc")
        return dw.permute(0, 3, 1, 2)
    raise ValueError(f"conv_wgrad: the NHWC weight-gradient kernel does not take geometry {geo} (square kernel <= 4x4, channels multiples "
                     "of 8, operands below 2 GiB); the staged chain of rounds 2-4 left the library in round 6")


def conv_wgrad_split(xs, gs, kernel_size, stride=1, padding=0, dilation=1, out=None):
    """fp32-grade weight gradient from split operands: xs = (x_hi, x_lo), gs = (g_hi, g_lo) bf16 channels-last ->
    dW (Cout,Cin,k,k) fp32 in channels-last memory (omnihd_conv_wgrad_split: one staging pass, one three-term GEMM launch).
    ``out``: an fp32 (Cout,Cin,k,k) tensor in channels_last memory to write into (a DDP reducer's view of the gradient inside
    its bucket: no copy afterwards); the returned tensor then aliases it."""
    for t in (*xs, *gs):
        _want_cl(t, "operand plane")
    B, cin, H, W = xs[0].shape
    _, cout, Ho, Wo = gs[0].shape
    k = int(kernel_size)
    dev = xs[0].device
    if out is not None:
        dw = out.permute(0, 2, 3, 1)
        if not (out.dtype == torch.float32 and tuple(out.shape) == (cout, cin, k, k) and dw.is_contiguous() and out.device == dev):
            raise ValueError("conv_wgrad_split: `out` must be an fp32 (Cout,Cin,k,k) tensor in channels_last memory on the operands' device")
    else:
        dw = torch.empty((cout, k, k, cin), dtype=torch.float32, device=dev)
    geo = (B, H, W, cin, Ho, Wo, cout, k, k, int(stride), int(padding), int(dilation))
    L = lib()
    if wgrad_nhwc_preferred(B, H, W, cin, Ho, Wo, cout, k, int(stride), int(padding), int(dilation)):
        # straight from the NHWC planes (csrc/conv_wgrad_nhwc.hip): no staging launches — the small and middle-sized layers
        g11 = (B, H, W, cin, Ho, Wo, cout, k, int(stride), int(padding), int(dilation))
        with _on(dev):
            nbytes = _SIZE_CACHE.get(("nhwc",) + g11)
            if nbytes is None:
                nbytes = _SIZE_CACHE[("nhwc",) + g11] = L.omnihd_conv_wgrad_nhwc_workspace_bytes(*g11)
            ws = _wgrad_workspace(nbytes, dev)
            check(L.omnihd_conv_wgrad_nhwc(xs[0].data_ptr(), xs[1].data_ptr(), gs[0].data_ptr(), gs[1].data_ptr(), dw.data_ptr(), *g11,
                                           ws.data_ptr(), ws.numel(), _raw_stream()), "omnihd_conv_wgrad_nhwc")
        return dw.permute(0, 3, 1, 2)
    raise ValueError(f"conv_wgrad_split: the NHWC weight-gradient kernel does not take geometry {geo}")


def wgrad_nhwc_preferred(B, H, W, cin, Ho, Wo, cout, k, stride, padding, dilation):
    """Does this library's weight-gradient kernel (csrc/conv_wgrad_nhwc.hip) take a geometry?  A RULE, not a measurement, so that a
    run's kernels — and the last bits of its gradients — never depend on timing noise: square kernels up to 4x4, channel counts
    multiples of 8, operands below 2 GiB.  (The staged chain of rounds 2-4 — k_to_kmajor + k_wgrad_shift / k_wgrad_split3,
    csrc/conv_wgrad.hip — left the library in round 6: scripts/lab/records/conv_wgrad_staged_chain.hip.txt; what the kernel does not
    take goes to MIOpen.)"""
    if k > 4:
        return False
    key = (B, H, W, cin, Ho, Wo, cout, k, stride, padding, dilation)
    hit = _NHWC_OK.get(key)
    if hit is None:
        hit = _NHWC_OK[key] = bool(lib().omnihd_conv_wgrad_nhwc_workspace_bytes(*key))
    return hit


_NHWC_OK = {}


def conv3x3_wgrad(x, grad_out):
    """3x3 / stride 1 / pad 1."""
    return conv_wgrad(x, grad_out, 3, 1, 1, 1)


def conv1x1_wgrad(x, grad_out):
    """1x1 / stride 1 (a strided 1x1 conv may pass the sub-sampled input)."""
    return conv_wgrad(x, grad_out, 1, 1, 0, 1)


def _pair_same(v):
    v = tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    return v[0] if len(v) == 2 and v[0] == v[1] else None


def conv_wgrad_supported(x, weight, stride, padding, dilation=(1, 1)):
    """bf16 device activations, square 1x1 / 3x3 kernel, equal stride / padding / dilation in both directions,
    channel counts multiples of 8 (the kernel pads them to 128 internally)."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4):
        return False
    if weight.shape[0] % 8 or weight.shape[1] % 8 or weight.shape[1] != x.shape[1]:
        return False
    k = tuple(weight.shape[2:])
    s, p, d = _pair_same(stride), _pair_same(padding), _pair_same(dilation)
    if k not in ((1, 1), (3, 3)) or s is None or p is None or d is None:
        return False
    if not (s >= 1 and d >= 1):
        return False
    B, cin, H, W = x.shape
    Ho, Wo = (H + 2 * p - d * (k[0] - 1) - 1) // s + 1, (W + 2 * p - d * (k[0] - 1) - 1) // s + 1
    return Ho > 0 and Wo > 0 and wgrad_nhwc_preferred(B, H, W, cin, Ho, Wo, weight.shape[0], k[0], s, p, d)


def conv3x3_wgrad_supported(x, weight):
    return conv_wgrad_supported(x, weight, (1, 1), (1, 1))


def conv_fwd_supported(x_shape, cout, k, stride, padding, dilation):
    """Geometries the implicit-GEMM forward / data-gradient kernel takes: stride 1, 'same' padding, k in {1,3},
    Cin a multiple of 64, Cout a multiple of 8."""
    B, cin, H, W = x_shape
    return (stride == 1 and k in (1, 3) and padding == dilation * (k // 2) and
            bool(lib().omnihd_conv_fwd_supported(B, H, W, cin, cout, k, dilation)))


# bench.py sets CONV_TIMING to a list and CONV_TIMING_GEOMETRY to one (cin, cout, k, H, W): HIP events are then recorded around
# every launch of our implicit-GEMM convolution kernels with that geometry inside the training step (forward and data gradient,
# bf16 and split forms) -> the in-step duration the `conv_roofline` block of the bench line is computed from
CONV_TIMING = None
CONV_TIMING_GEOMETRY = None


def _conv_timed(kind, geo, launch):
    if CONV_TIMING is None or geo != CONV_TIMING_GEOMETRY:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    CONV_TIMING.append((kind, e0, e1))


def conv_fwd(x, w_cl, bias=None, dilation=1, tile=0):
    """y = conv2d(x, w, bias, stride 1, padding = dilation*(k//2)) on the matrix cores (csrc/conv_igemm.hip).
    x (B,Cin,H,W) bf16 channels-last, w_cl (Cout,Cin,k,k) bf16 in channels_last memory format ((Cout,k,k,Cin) memory),
    bias (Cout,) fp32 or None -> (B,Cout,H,W) bf16 channels-last."""
    _want_cl(x, "x")
    if w_cl.dtype != torch.bfloat16 or w_cl.dim() != 4 or not w_cl.is_contiguous(memory_format=torch.channels_last):
        raise TypeError("w must be a 4-D bf16 tensor in channels_last memory format")
    B, cin, H, W = x.shape
    cout, k = w_cl.shape[0], w_cl.shape[2]
    # (B,Cout,H,W) with channels-last strides = NHWC memory; not a view of anything (a custom Function must not hand out views)
    y = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    with _on(x.device):
        _conv_timed("bf16", (cin, cout, k, H, W), lambda: check(
            lib().omnihd_conv_fwd_bf16(x.data_ptr(), w_cl.data_ptr(), None if bias is None else _f32c(bias).data_ptr(),
                                       y.data_ptr(), B, H, W, cin, cout, k, int(dilation), int(tile), _raw_stream()),
            "omnihd_conv_fwd_bf16"))
    return y


def conv_dgrad_weights(w_cl, out=None):
    """(Cout,Cin,k,k) channels_last bf16 -> (Cin,Cout,k,k) channels_last bf16 with mirrored taps: the weights with which
    the data gradient is ``conv_fwd(grad_out, wt)``."""
    cout, cin, k, _ = w_cl.shape
    wt = out if out is not None else torch.empty((cin, cout, k, k), dtype=torch.bfloat16, device=w_cl.device,
                                                 memory_format=torch.channels_last)
    with _on(w_cl.device):
        check(lib().omnihd_conv_dgrad_weights(w_cl.data_ptr(), wt.data_ptr(), cout, cin, k, _raw_stream()),
              "omnihd_conv_dgrad_weights")
    return wt


_GEN_OK = {}


def _conv_out_hw(H, W, k, s, p, d):
    return (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1


def conv_gen_supported(mode, x_shape, cout, k, stride, padding, dilation):
    """The general implicit-GEMM kernel (csrc/conv_gen.hip) takes this pass of conv2d(x (B,Cin,H,W), w (Cout,Cin,k,k), stride,
    padding, dilation): mode 0 = forward, mode 1 = data gradient (strided forms included; OMNIHD_CONV_GEN=0 turns it off)."""
    if _env("OMNIHD_CONV_GEN", "1") == "0":
        return False
    key = (int(mode), tuple(x_shape), int(cout), int(k), int(stride), int(padding), int(dilation))
    hit = _GEN_OK.get(key)
    if hit is None:
        B, cin, H, W = x_shape
        s, p, d = int(stride), int(padding), int(dilation)
        if s < 1 or H + 2 * p - d * (k - 1) - 1 < 0 or W + 2 * p - d * (k - 1) - 1 < 0:
            hit = False
        else:
            Ho, Wo = _conv_out_hw(H, W, k, s, p, d)
            hit = bool(lib().omnihd_conv_gen_supported(int(mode), B, H, W, cin, Ho, Wo, int(cout), int(k), s, p, d))
        if len(_GEN_OK) > 4096:
            _GEN_OK.clear()
        _GEN_OK[key] = hit
    return hit


def conv_gen(mode, src, w, bias, x_shape, cout, k, stride, padding, dilation):
    """One pass of conv2d(x (B,Cin,H,W), w (Cout,Cin,k,k), stride, padding, dilation) on the general implicit-GEMM kernel:
      mode 0: src = x,    w = weight image in (Cout,k,k,Cin) memory          -> y  (B,Cout,Ho,Wo)   (+ fp32 bias)
      mode 1: src = gout, w = data-gradient image ((Cin,k,k,Cout), mirrored) -> gx (B,Cin,H,W), every pixel written once
    ``src`` / ``w``: bf16 channels-last tensors (bf16 result) or (hi, lo) pairs of them (fp32-grade split form, fp32 result)."""
    split = isinstance(src, (tuple, list))
    s0 = src[0] if split else src
    w0 = w[0] if split else w
    B, cin, H, W = x_shape
    s_, p_, d_ = int(stride), int(padding), int(dilation)
    Ho, Wo = _conv_out_hw(H, W, k, s_, p_, d_)
    want = (B, cin, H, W) if mode == 0 else (B, cout, Ho, Wo)
    for t in (tuple(src) if split else (src,)):
        _want_cl(t, "source")
        if tuple(t.shape) != want:
            raise ValueError(f"conv_gen: source {tuple(t.shape)}, the pass reads {want}")
    for t in (tuple(w) if split else (w,)):
        if t.dtype != torch.bfloat16 or t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
            raise TypeError("weight images must be 4-D bf16 tensors in channels_last memory format")
    out_shape = (B, cout, Ho, Wo) if mode == 0 else (B, cin, H, W)
    y = torch.empty(out_shape, dtype=torch.float32 if split else torch.bfloat16, device=s0.device, memory_format=torch.channels_last)
    with _on(y.device):
        check(lib().omnihd_conv_gen(int(mode), s0.data_ptr(), src[1].data_ptr() if split else None, w0.data_ptr(),
                                    w[1].data_ptr() if split else None, None if bias is None else _f32c(bias).data_ptr(), y.data_ptr(),
                                    B, H, W, cin, Ho, Wo, int(cout), int(k), s_, p_, d_, _raw_stream()), "omnihd_conv_gen")
    return y


def deterministic():
    """OMNIHD_DETERMINISTIC=1: every convolution pass this library has a kernel for runs on it (no per-geometry race against the
    library kernels, whose fp32 solvers for strided layers and small weight gradients accumulate with atomics), so that a
    training step is run-to-run identical bit for bit (tests/test_determinism_gpu.py)."""
    return _env("OMNIHD_DETERMINISTIC", "0") == "1"


# Which implementation computes the weight gradient of a given convolution geometry: the MFMA kernel chain of
# this library ("hip") or MIOpen ("miopen").  The staged GEMM wins by 2-4x on the BEV-sized convolutions and on
# small feature maps, MIOpen's direct implicit GEMM wins where the pixel axis is long and the channel counts are
# small (scripts/wgrad_census.py), and which of MIOpen's solvers is picked depends on the box — so, like MIOpen's
# own find step, the choice is MEASURED once per geometry (a handful of launches during warm-up) and cached.
# OMNIHD_WGRAD_POLICY = tune (default) | hip | miopen.  Measured in the full R1 step with MIOpen in find mode
# (torch.backends.cudnn.benchmark = True): tune 35.8 ms (our chain on the 8 BEV-sized geometries, MIOpen on 37), hip 38.5 ms,
# miopen 41.6 ms.  With MIOpen's immediate-mode kernels (benchmark off) our chain wins nearly everywhere: 45.4 vs 46.3 ms.
class _ChoiceTable(dict):
    """{geometry + (device index,): implementation}.  A lookup that misses falls back to the PERSISTED table of this name
    (omnihd-scenes_amd/kernel_choices/gfx950.json: the choices measured once on an MI355X and committed, keyed without the
    device index) before anything is measured, so a run's kernels — hence its numerics, launch count and speed — do not
    depend on the timing noise of its first steps.  A geometry the file does not know is measured as before and counted as
    a miss (``choice_table_info()``); ``save_choice_table`` writes the merged table back."""

    def __init__(self, name):
        super().__init__()
        self.name = name

    def get(self, key, default=None):
        if key in self:
            return self[key]
        hit = _persisted_choices().get(self.name, {}).get(tuple(key[:-1]))
        if hit is not None:
            self[key] = hit
            return hit
        return default

    def measured(self, key, value):
        """Record a choice that had to be measured in this process (a miss of the persisted table)."""
        self[key] = value
        _CHOICE_INFO["misses"] += 1
        return value


_CHOICE_INFO = {"path": None, "sha256": None, "entries": 0, "misses": 0, "loaded": False}
_PERSISTED = {}


def _tuplify(x):
    return tuple(_tuplify(v) for v in x) if isinstance(x, list) else x


def _persisted_choices():
    if not _CHOICE_INFO["loaded"]:
        _CHOICE_INFO["loaded"] = True
        path = _env("OMNIHD_CHOICE_TABLE")
        if path is None:
            path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kernel_choices", "gfx950.json")
        if path and path != "off" and os.path.exists(path):
            import hashlib
            import json
            raw = open(path, "rb").read()
            doc = json.loads(raw)
            # The winners are properties of ONE chip generation and of the MIOpen mode they were measured against (find-mode
            # kernels beat ours on geometries where the immediate-mode ones lose).  A table captured on another architecture is
            # ignored (everything is measured, every geometry a miss); a table captured in the other MIOpen mode is still used —
            # a run's kernels stay reproducible — unless OMNIHD_CHOICE_TABLE_STRICT=1, and the mismatch is reported.
            want_arch = str(doc.get("arch", "gfx950"))
            arch = _device_arch()
            arch_ok = arch is None or arch.split(":")[0] == want_arch
            find_now = bool(torch.backends.cudnn.benchmark)
            mode_ok = bool(doc.get("miopen_find", True)) == find_now
            _CHOICE_INFO.update(arch=arch, table_arch=want_arch, miopen_find=find_now, table_miopen_find=bool(doc.get("miopen_find", True)))
            if arch_ok and (mode_ok or _env("OMNIHD_CHOICE_TABLE_STRICT", "0") != "1"):
                for name in ("conv", "wgrad", "split"):
                    _PERSISTED[name] = {_tuplify(json.loads(k)): v for k, v in doc.get(name, {}).items()}
                _CHOICE_INFO.update(path=path, sha256=hashlib.sha256(raw).hexdigest(), entries=sum(len(v) for v in _PERSISTED.values()))
            else:
                _CHOICE_INFO.update(path=path, sha256=hashlib.sha256(raw).hexdigest(), entries=0,
                                    ignored="architecture" if not arch_ok else "miopen mode (OMNIHD_CHOICE_TABLE_STRICT=1)")
    return _PERSISTED


def _device_arch():
    try:
        if torch.cuda.is_available():
            return str(torch.cuda.get_device_properties(torch.cuda.current_device()).gcnArchName)
    except Exception:
        pass
    return None


def choice_table_info():
    """{'path', 'sha256', 'entries', 'misses'} of the persisted kernel-choice table as this process sees it (``misses`` =
    geometries that had to be measured here because the table did not hold them)."""
    _persisted_choices()
    info = {k: _CHOICE_INFO[k] for k in ("path", "sha256", "entries", "misses")}
    for k in ("arch", "table_arch", "miopen_find", "table_miopen_find", "ignored"):
        if k in _CHOICE_INFO:
            info[k] = _CHOICE_INFO[k]
    if info["path"]:
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        if os.path.abspath(info["path"]).startswith(root + os.sep):
            info["path"] = os.path.relpath(info["path"], root)          # as committed, not where this checkout happens to live
    return info


def save_choice_table(path, note=""):
    """Write every choice known to this process (persisted + measured) as the table ``_ChoiceTable`` reads; returns the count."""
    import json
    doc = {"note": note or "kernel choices per convolution geometry, measured on MI355X (gfx950); keys = geometry tuples without the device index",
           "arch": (_device_arch() or "gfx950").split(":")[0], "miopen_find": bool(torch.backends.cudnn.benchmark)}
    n = 0
    for name, table in (("conv", _CONV_CHOICE), ("wgrad", _WGRAD_CHOICE), ("split", _SPLIT_CHOICE)):
        merged = dict(_persisted_choices().get(name, {}))
        merged.update({tuple(k[:-1]): v for k, v in table.items()})
        doc[name] = {json.dumps(list(k)): v for k, v in sorted(merged.items(), key=lambda kv: json.dumps(list(kv[0])))}
        n += len(merged)
    with open(path, "w") as f:
        json.dump(doc, f, indent=0, sort_keys=True)
        f.write("\n")
    return n


_WGRAD_CHOICE = _ChoiceTable("wgrad")


def _miopen_wgrad(x, g, weight, stride, padding, dilation):
    return torch.ops.aten.convolution_backward(g, x, weight, None, stride, padding, dilation, False, [0, 0], 1,
                                               [False, True, False])[1]


def _tuned_wgrad(x, g, weight, stride, padding, dilation):
    policy = "hip" if deterministic() else _env("OMNIHD_WGRAD_POLICY", "tune")
    k = weight.shape[2]
    run_hip = lambda: conv_wgrad(x, g, k, stride[0], padding[0], dilation[0])
    if policy == "hip":
        return run_hip()
    run_mi = lambda: _miopen_wgrad(x, g, weight, stride, padding, dilation)
    if policy == "miopen":
        return run_mi()
    key = (tuple(x.shape), g.shape[1], k, stride[0], padding[0], dilation[0], x.device.index)
    choice = _WGRAD_CHOICE.get(key)
    if choice is None:
        def clock(fn):
            return _clock(fn, x.device, n=5, warm=2)
        choice = _WGRAD_CHOICE.measured(key, "hip" if clock(run_hip) <= clock(run_mi) else "miopen")
    return run_hip() if choice == "hip" else run_mi()


def wgrad_choice_for(x_shape, cout, k, stride, padding, dilation, device_index):
    """'hip' | 'miopen' | None (not measured yet) for a convolution geometry under the current policy."""
    policy = "hip" if deterministic() else _env("OMNIHD_WGRAD_POLICY", "tune")
    if policy != "tune":
        return policy
    return _WGRAD_CHOICE.get((tuple(x_shape), cout, k, stride, padding, dilation, device_index))


def conv_all_miopen(x_shape, cout, k, stride, padding, dilation, device_index):
    """True once EVERY direction of a convolution geometry has been measured in MIOpen's favour (or cannot run here):
    the layer is then a plain torch convolution again (no Python in its backward)."""
    if wgrad_choice_for(x_shape, cout, k, stride, padding, dilation, device_index) != "miopen":
        return False
    if _conv_policy() == "miopen":
        return True
    if _conv_policy() == "hip":
        return False
    B, cin, H, W = x_shape
    same = stride == 1 and k in (1, 3) and padding == dilation * (k // 2)
    if same and cin % 64 == 0 and cout % 8 == 0:
        if _CONV_CHOICE.get(("fwd", tuple(x_shape), cout, k, dilation, device_index)) != "miopen":
            return False
    elif cout % 8 == 0 and k <= 4 and conv_gen_supported(0, tuple(x_shape), cout, k, stride, padding, dilation):
        if _CONV_CHOICE.get(("fwd_gen", tuple(x_shape), cout, k, stride, padding, dilation, device_index)) != "miopen":
            return False
    if same and cout % 64 == 0 and cin % 8 == 0:
        if _CONV_CHOICE.get(("dgrad", (B, cout, H, W), cin, k, dilation, device_index)) != "miopen":
            return False
    elif cin % 8 == 0 and cout % 8 == 0 and k in (1, 3) and conv_gen_supported(1, tuple(x_shape), cout, k, stride, padding, dilation):
        if _CONV_CHOICE.get(("dgrad_gen", tuple(x_shape), cout, k, stride, padding, dilation, device_index)) != "miopen":
            return False
    return True


def wgrad_choices():
    """{geometry: 'hip' | 'miopen'} decided so far (for logs and DESIGN.md tables)."""
    return dict(_WGRAD_CHOICE)


# bf16 images of the fp32 master weights.  Each convolution needs its weight rounded to bf16 once per step; done
# layer by layer that is ~100 tiny cast kernels (and their Python) per step.  The images are kept here, keyed by
# the parameter, and reused while the parameter's version counter is unchanged; a training loop may refresh all
# of them with ONE multi-tensor copy right after the optimiser step (``refresh_bf16_shadows``).

# When is a cached image of a master weight stale?  The parameter's autograd version counter moves on every in-place write that
# goes through torch's dispatcher (copy_, load_state_dict, foreach optimisers) — but NOT on the fused optimisers
# (torch.optim.AdamW(fused=True) updates the parameters inside one kernel and leaves ``_version`` alone; found in round 3: the
# caches below then served step-0 images for ever).  So every optimiser step of ANY optimiser also moves a global generation
# counter (torch's global step post-hook), and an image is current only if both match.
_WEIGHT_GEN = [0]


def _bump_weight_generation(*_a, **_k):
    _WEIGHT_GEN[0] += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_step_hook
    _WEIGHT_GEN_HOOK = _reg_step_hook(_bump_weight_generation)
except Exception:            # pragma: no cover - very old torch: callers must use weights_changed()
    _WEIGHT_GEN_HOOK = None


def weights_changed():
    """Tell the weight-image caches that parameters were changed in a way that moves neither their version counter nor an
    optimiser step (writes through ``param.data``, custom kernels on the raw pointer)."""
    _bump_weight_generation()


def _wver(w):
    # (frozen weights are not touched by an optimiser: only the version counter applies to them)
    return (w._version, _WEIGHT_GEN[0] if w.requires_grad else -1)


_BF16_SHADOW = {}
_SHADOW_EPOCH = [0]          # moves whenever an image BUFFER is created or replaced (the cached refresh plan holds raw pointers)


def bf16_of(weight):
    if weight.dtype == torch.bfloat16:
        return weight.detach()
    e = _BF16_SHADOW.get(id(weight))
    if e is not None and e[0]() is weight and e[1] == _wver(weight) and e[2].device == weight.device:
        return e[2]
    # convolution weights: the image lives in channels_last memory ((Cout,k,k,Cin), what the implicit-GEMM kernels and the
    # NHWC library kernels read), so no layer pays a layout copy per step
    if weight.dim() == 4 and e is not None and e[0]() is weight and e[2].shape == weight.shape and e[2].device == weight.device:
        shadow = e[2]
        shadow.copy_(weight.detach())
    elif weight.dim() == 4:
        shadow = weight.detach().to(torch.bfloat16, memory_format=torch.channels_last)
        _SHADOW_EPOCH[0] += 1
    else:
        shadow = weight.detach().to(torch.bfloat16)
        _SHADOW_EPOCH[0] += 1
    _BF16_SHADOW[id(weight)] = (weakref.ref(weight), _wver(weight), shadow)
    return shadow


_BF16_DGRAD = {}


def bf16_dgrad_image(param, wb):
    """The bf16 weights re-laid for the data gradient ((Cin,k,k,Cout) memory, taps mirrored), cached per parameter version and
    refreshed together with the bf16 image by ``refresh_bf16_shadows``.  ``param`` None: not cached."""
    w_cl = wb if wb.is_contiguous(memory_format=torch.channels_last) else wb.contiguous(memory_format=torch.channels_last)
    if param is None:
        return conv_dgrad_weights(w_cl)
    e = _BF16_DGRAD.get(id(param))
    if e is not None and e[0]() is param and e[2].device == wb.device and e[2].shape[:2] == (wb.shape[1], wb.shape[0]):
        if e[1] == _wver(param):
            return e[2]
        img = conv_dgrad_weights(w_cl, out=e[2])
    else:
        img = conv_dgrad_weights(w_cl)
        _SHADOW_EPOCH[0] += 1
    if len(_BF16_DGRAD) > 4096:
        for k in [k for k, v in _BF16_DGRAD.items() if v[0]() is None]:
            del _BF16_DGRAD[k]
    _BF16_DGRAD[id(param)] = (weakref.ref(param), _wver(param), img)
    return img


class _Bf16Weight(torch.autograd.Function):
    """The cached bf16 image of an fp32 master weight as a differentiable function of it (gradient cast back)."""

    @staticmethod
    def forward(ctx, weight):
        ctx.wdtype = weight.dtype
        return bf16_of(weight).view_as(weight)          # a fresh alias: the cached tensor itself must not get a grad_fn

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.wdtype)


def bf16_weight(weight):
    if weight.dtype == torch.bfloat16 or not weight.is_cuda:
        return weight
    return _Bf16Weight.apply(weight) if weight.requires_grad and torch.is_grad_enabled() else bf16_of(weight)


_BF16_PLAN = {}


def refresh_bf16_shadows():
    """Bring the bf16 images of all TRAINABLE weights up to date: convolution weights (and their data-gradient images, where a layer
    has one) with ONE launch of omnihd_weight_images, everything else with one fused copy; returns how many were refreshed.  The
    launch plan (device table, copy lists) is kept while the set of registered images is unchanged, so a steady-state call is a
    table lookup, two launches and one pass over the entries to stamp them current."""
    sig = (len(_BF16_SHADOW), len(_BF16_DGRAD), _SHADOW_EPOCH[0])
    plan = _BF16_PLAN.get("plan")

    def moved(ref, where):
        # a Parameter may keep its identity and get NEW storage (module.to(memory_format=...), ``param.data = ema``,
        # vector_to_parameters, sharded optimisers): neither its version counter nor the signature above moves, while the plan
        # holds the old storage's raw pointer and strides (or a detached alias of it)
        w = ref()
        return w is None or (w.data_ptr(), w.stride()) != where
    if plan is None or plan[0] != sig or any(moved(ref, where) for _k, ref, _s, _d, where in plan[1]):
        entries, per_dev, src, dst = [], {}, [], []
        for k, (ref, ver, shadow) in list(_BF16_SHADOW.items()):
            w = ref()
            if w is None:
                del _BF16_SHADOW[k]
                _BF16_DGRAD.pop(k, None)
                continue
            if not w.requires_grad or w.device != shadow.device or w.shape != shadow.shape:
                continue                          # frozen weights change only through torch (version counter): bf16_of sees that
            if (w.dim() == 4 and w.dtype == torch.float32 and w.is_cuda and w.shape[2] == w.shape[3] and 1 <= w.shape[2] <= 4
                    and shadow.is_contiguous(memory_format=torch.channels_last)):
                d = _BF16_DGRAD.get(k)
                d = d[2] if d is not None and d[0]() is w and d[2].device == w.device else None
                cout, cin, kk, _ = w.shape
                per_dev.setdefault(w.device, []).append((w.data_ptr(),) + tuple(w.stride()) + (shadow.data_ptr(), 0, 0 if d is None else d.data_ptr(),
                                                                                               0, cout, cin, kk))
                entries.append((k, ref, shadow, d, (w.data_ptr(), w.stride())))
            else:
                src.append(w.detach()); dst.append(shadow); entries.append((k, ref, shadow, None, (w.data_ptr(), w.stride())))
        plan = _BF16_PLAN["plan"] = ((len(_BF16_SHADOW), len(_BF16_DGRAD), _SHADOW_EPOCH[0]), entries, per_dev, src, dst)
    _sig, entries, per_dev, src, dst = plan
    for dev, recs in per_dev.items():
        weight_images(recs, dev)
    if src:
        torch._foreach_copy_(dst, src)
    for k, ref, shadow, d, _where in entries:
        ver = _wver(ref())
        _BF16_SHADOW[k] = (ref, ver, shadow)
        if d is not None:
            _BF16_DGRAD[k] = (ref, ver, d)
    return len(entries)


# Forward and data gradient of the stride-1 "same" convolutions: the implicit-GEMM MFMA kernel of this library
# (csrc/conv_igemm.hip, two tile shapes) or MIOpen — like the weight gradient, a measured choice per geometry and direction
# (OMNIHD_CONV_POLICY = tune (default) | hip | miopen).  Measured at the BEV sizes (scripts/lab/conv_bench.py): the data
# gradient is ours on every 3x3 geometry (854 vs 700 TFLOP/s on 1024->1024 at 160x240), the forward is a close race
# (861 vs 875-966), 1x1 convolutions stay on MIOpen.
_CONV_CHOICE = _ChoiceTable("conv")
_CONV_IMPLS = ("hip", "hip128x256", "miopen")


def _conv_policy():
    return "hip" if deterministic() else _env("OMNIHD_CONV_POLICY", "tune")


def _conv_impl(direction, x, w_cl, stride, padding, dilation, run_miopen, bias=None, n_out=None, k=None, in_shape=None):
    """Run one direction ('fwd': x = input, w_cl = weights; 'dgrad': x = grad_out, w_cl = data-gradient weights) with the
    implementation chosen for its geometry.  ``w_cl`` may be a function returning the weights (with ``n_out`` = their
    output channels and ``k``): the data gradient's mirrored / transposed weights are then only made when our kernel runs.
    ``in_shape``: the convolution's INPUT shape (B,Cin,H,W) — with it, geometries the stride-1 kernels do not take (strides,
    other paddings, channel counts that are multiples of 8 only) run on the general kernel of csrc/conv_gen.hip."""
    lazy = callable(w_cl)
    weights = (lambda: w_cl()) if lazy else (lambda: w_cl)
    if not lazy:
        n_out, k = w_cl.shape[0], w_cl.shape[2]
    policy = _conv_policy()
    square = stride[0] == stride[1] and padding[0] == padding[1] and dilation[0] == dilation[1]
    ours = (x.dtype == torch.bfloat16 and square and conv_fwd_supported(x.shape, n_out, k, stride[0], padding[0], dilation[0]))
    if not ours and policy != "miopen" and in_shape is not None and x.dtype == torch.bfloat16 and square and k <= 4:
        mode = 0 if direction == "fwd" else 1
        cout_conv = n_out if mode == 0 else x.shape[1]
        if n_out % 8 == 0 and conv_gen_supported(mode, tuple(in_shape), cout_conv, k, stride[0], padding[0], dilation[0]):
            run_gen = lambda: conv_gen(mode, x, weights(), bias, tuple(in_shape), cout_conv, k, stride[0], padding[0], dilation[0])
            if policy == "hip":
                return run_gen()
            key = (direction + "_gen", tuple(in_shape), cout_conv, k, stride[0], padding[0], dilation[0], x.device.index)
            choice = _CONV_CHOICE.get(key)
            if choice is None:
                clock = lambda fn: _clock(fn, x.device, n=5, warm=2)
                choice = _CONV_CHOICE.measured(key, "hip" if clock(run_gen) <= clock(run_miopen) else "miopen")
            return run_gen() if choice == "hip" else run_miopen()
    if not ours or policy == "miopen":
        return run_miopen()
    made = []

    def run_hip(tile, timing=False):                        # fp32 bias added before the one rounding
        if timing and lazy:                                 # the measurement pays for the weight transform every time
            return conv_fwd(x, weights(), bias, dilation[0], tile)
        if not made:
            made.append(weights())
        return conv_fwd(x, made[0], bias, dilation[0], tile)
    if policy == "hip":
        return run_hip(0)
    key = (direction, tuple(x.shape), n_out, k, dilation[0], x.device.index)
    choice = _CONV_CHOICE.get(key)
    if choice is None:
        def clock(fn):
            return _clock(fn, x.device, n=5, warm=2)
        # "hip": the library's own pick (3x3 with dilation <= 8 at BEV sizes: the row-shift kernel, else the 256x128 tile,
        # 128x128 for small problems); "hip128x256": the wide-N tile
        times = {"hip": clock(lambda: run_hip(0, True)), "hip128x256": clock(lambda: run_hip(254, True)), "miopen": clock(run_miopen)}
        choice = _CONV_CHOICE.measured(key, min(times, key=times.get))
    if choice == "miopen":
        return run_miopen()
    return run_hip(0 if choice == "hip" else 254)


def conv_choices():
    """{(direction, geometry...): implementation} decided so far (for logs and DESIGN.md tables)."""
    return dict(_CONV_CHOICE)


def sync_tuned_choices(group=None, src=0):
    """Make every rank of ``group`` use rank ``src``'s measured kernel choices (convolution forward / data gradient / weight
    gradient per geometry).  The measurements run inside forward / backward on each rank separately and a noisy one could put
    two ranks on kernels with different bf16 summation orders; call this once after the set-up steps (the harness does) from
    the thread that owns the process group, outside forward / backward.  Device indices in the keys are mapped to the local
    device.  Returns the number of entries that changed on this rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    me = _current_device() if torch.cuda.is_available() else None
    strip = lambda table: {k[:-1]: v for k, v in table.items()}
    payload = [(strip(_CONV_CHOICE), strip(_WGRAD_CHOICE), strip(_SPLIT_CHOICE))] if dist.get_rank(group) == src else [None]
    dist.broadcast_object_list(payload, src=src, group=group)
    changed = 0
    for table, theirs in ((_CONV_CHOICE, payload[0][0]), (_WGRAD_CHOICE, payload[0][1]), (_SPLIT_CHOICE, payload[0][2])):
        for k, v in theirs.items():
            if table.get(k + (me,)) != v:
                table[k + (me,)] = v
                changed += 1
    return changed


# --------------------------------------------------------------------------------------------
# fp32-grade convolutions on the bf16 matrix cores: 3-term split (hi*hi + hi*lo + lo*hi), fp32 accumulation
# --------------------------------------------------------------------------------------------
def split_f32(t):
    """fp32 tensor (dense in its memory format) -> (hi, lo) bf16 tensors of the same shape and strides with
    t = hi + lo up to 2^-17 |t| (omnihd_split_f32)."""
    if not (t.is_cuda and t.dtype == torch.float32):
        raise TypeError("split_f32 takes an fp32 CUDA(HIP) tensor")
    if not (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))):
        t = t.contiguous()
    # both planes in ONE allocation, back to back (the row-shift kernels reach them through one buffer descriptor)
    n = t.numel()
    pitch = (n + 7) // 8 * 8                                  # 16-byte aligned planes
    planes = torch.empty(2 * pitch, dtype=torch.bfloat16, device=t.device)
    hi, lo = (planes[i * pitch:i * pitch + n].as_strided(t.shape, t.stride()) for i in (0, 1))
    with _on(t.device):
        check(lib().omnihd_split_f32(t.data_ptr(), t.numel(), hi.data_ptr(), lo.data_ptr(), _raw_stream()), "omnihd_split_f32")
    return hi, lo


def _alloc_planes(t):
    """Uninitialised (hi, lo) planes for ``t`` in split_f32's layout (one allocation, lo at a 16-byte-aligned pitch behind hi)."""
    n = t.numel()
    pitch = (n + 7) // 8 * 8
    planes = torch.empty(2 * pitch, dtype=torch.bfloat16, device=t.device)
    return tuple(planes[i * pitch:i * pitch + n].as_strided(t.shape, t.stride()) for i in (0, 1))


# Planes handed from a producer kernel to the split convolution that reads the tensor next (round 3): the fused BatchNorm kernels
# can write the two bf16 planes of their fp32 output in the same pass (omnihd_bn_train_fwd_f32_planes / ..._bwd_f32_planes), which
# saves the convolution's own split pass (a read + write of the whole tensor and a launch).  Protocol: the producer tags its
# output tensor OBJECT with (planes, version counter, producer key); the convolution uses the planes if the tag is there and the
# tensor has not been written since; on a miss it notes the producer key, and from the next step on that producer writes planes.
# A producer whose planes nobody picked up stops writing them.  Measured in the R1 fp32 step (alternating runs of
# scripts/lab/step_times.py): 50.9 ms with, 50.6 ms without — the BatchNorm kernels' extra 4 B/element of stores cost what the
# convolutions' split passes saved, so it is OFF by default (OMNIHD_SPLIT_HANDOVER=1 turns it on; results are bit-identical).
_PLANES_WANTED = set()
_PLANES_UNUSED = {}
HANDOVER_STATS = {"taken": 0, "stale": 0, "asked": 0, "untagged": 0}


def planes_wanted(key):
    return key in _PLANES_WANTED and _env("OMNIHD_SPLIT_HANDOVER", "0") == "1"


def tag_planes(t, planes, key):
    t._omnihd_planes = (planes, t._version, key)
    n = _PLANES_UNUSED.get(key, 0) + 1
    _PLANES_UNUSED[key] = n
    if n > 8:                                      # eight tensors in a row that no convolution took: stop producing
        _PLANES_WANTED.discard(key)
        _PLANES_UNUSED[key] = 0


def tag_producer(t, key):
    t._omnihd_planes = (None, t._version, key)


# The TF32-grade form (OMNIHD_FP32_CONV=f16) uses the same tags with ONE plane: the IEEE half of the tensor, written by the producer's
# epilogue instead of a cast pass (2 bytes per element written there against 4 read + 2 written here, and a launch less per layer:
# ON whenever the policy is f16).  Its gradients travel as fp32 with the amax the producer's backward accumulated (``_omnihd_amax``).
_HALF_WANTED = set()


def f16_handover():
    """TF32-grade policy with the producers' hand-over on (OMNIHD_F16_HANDOVER=0: every convolution runs its own cast / amax passes —
    the A/B switch of tests/test_conv_f16_gpu.py and of DESIGN.md 4.6b's numbers)."""
    return _fp32_policy() == "f16" and _env("OMNIHD_F16_HANDOVER", "1") != "0"


def half_wanted(key):
    return key in _HALF_WANTED and f16_handover()


def tag_half(t, plane, key):
    t._omnihd_planes = ((plane,), t._version, key)
    n = _PLANES_UNUSED.get(key, 0) + 1
    _PLANES_UNUSED[key] = n
    if n > 8:                                      # eight tensors in a row that no TF32-grade convolution took: stop producing
        _HALF_WANTED.discard(key)
        _PLANES_UNUSED[key] = 0


def take_half(t):
    """The half plane a producer attached to ``t`` (fp32, dense, unmodified since), or None — in which case the producer, if there
    is one, is asked to write it from now on."""
    tag = getattr(t, "_omnihd_planes", None)
    if tag is None:
        return None
    planes, version, key = tag
    if planes is None or len(planes) != 1:
        _HALF_WANTED.add(key)
        return None
    if version != t._version or planes[0].shape != t.shape or planes[0].stride() != t.stride():
        HANDOVER_STATS["stale"] += 1
        return None
    _PLANES_UNUSED[key] = 0
    HANDOVER_STATS["taken_half"] = HANDOVER_STATS.get("taken_half", 0) + 1
    return planes[0]


def take_planes(t):
    """The planes a producer attached to ``t`` (fp32, channels_last-dense, unmodified since), or None — in which case the
    producer, if there is one, is asked to write them from now on."""
    tag = getattr(t, "_omnihd_planes", None)
    if tag is None:
        HANDOVER_STATS["untagged"] += 1
        return None
    planes, version, key = tag
    if planes is not None and len(planes) != 2:      # the half plane of the TF32-grade form: not ours
        return None
    if planes is None:
        if _env("OMNIHD_SPLIT_HANDOVER", "0") == "1":
            _PLANES_WANTED.add(key)
        HANDOVER_STATS["asked"] += 1
        return None
    if version != t._version or planes[0].shape != t.shape or planes[0].stride() != t.stride():
        HANDOVER_STATS["stale"] += 1
        return None
    _PLANES_UNUSED[key] = 0
    HANDOVER_STATS["taken"] += 1
    return planes


_SPLIT_SHADOW = {}


def split_dgrad_weights(ws):
    """(w_hi, w_lo) (Cout,Cin,k,k) channels_last -> the two planes re-laid for the data gradient ((Cin,Cout,k,k) channels_last,
    taps mirrored), in one allocation back to back (the kernels reach both planes through one buffer descriptor)."""
    hi, lo = ws
    cout, cin, k, _ = hi.shape
    both = torch.empty((2, cin, k, k, cout), dtype=torch.bfloat16, device=hi.device)
    return tuple(conv_dgrad_weights(p_, both[i].permute(0, 3, 1, 2)) for i, p_ in enumerate((hi, lo)))


def split_weight(weight, dgrad=False):
    """(hi, lo) bf16 planes of an fp32 convolution weight in channels_last memory ((Cout,k,k,Cin)), cached while the
    parameter's version is unchanged; ``dgrad=True``: the planes re-laid for the data gradient ((Cin,k,k,Cout), taps mirrored)."""
    if weight.grad_fn is not None:
        # a temporary computed from a parameter (the block-diagonal matrix DCN rebuilds every forward): never seen again under
        # this id, so caching it would only pin its planes until the 4096-entry sweep
        planes = split_f32(weight.detach().float().contiguous(memory_format=torch.channels_last))
        return split_dgrad_weights(planes) if dgrad else planes
    key = (id(weight), dgrad)
    e = _SPLIT_SHADOW.get(key)
    if e is not None and e[0]() is weight and e[1] == _wver(weight) and e[2][0].device == weight.device:
        return e[2]
    if dgrad:
        planes = split_dgrad_weights(split_weight(weight))
    else:
        planes = split_f32(weight.detach().float().contiguous(memory_format=torch.channels_last))
    if len(_SPLIT_SHADOW) > 4096:
        for k in [k for k, v in _SPLIT_SHADOW.items() if v[0]() is None]:
            del _SPLIT_SHADOW[k]
    _SPLIT_SHADOW[key] = (weakref.ref(weight), _wver(weight), planes)
    return planes


_WIMG_DTYPE = None
_WIMG_TABLES = {}
WIMG_STATS = {"hit": 0, "miss": 0}


def _weight_image_table(records, dev, per_tap=False):
    """Device table of omnihd_weight_images records (cached while the same buffers are asked for)."""
    global _WIMG_DTYPE
    import numpy as np
    if _WIMG_DTYPE is None:
        _WIMG_DTYPE = np.dtype([("src", "<u8"), ("so", "<i8"), ("si", "<i8"), ("sy", "<i8"), ("sx", "<i8"), ("f_hi", "<u8"), ("f_lo", "<u8"),
                                ("d_hi", "<u8"), ("d_lo", "<u8"), ("cout", "<i4"), ("cin", "<i4"), ("k", "<i4"), ("first_block", "<i4")])
    key = (dev.index, bool(per_tap), tuple(records))
    hit = _WIMG_TABLES.get(key)
    WIMG_STATS["hit" if hit is not None else "miss"] += 1        # (a miss is a BLOCKING host-to-device copy: fast_paths_report shows the count)
    if hit is None:
        arr = np.zeros(len(records), dtype=_WIMG_DTYPE)
        first = 0
        for n, r in enumerate(records):
            arr[n] = r + (first,)
            if per_tap:                                   # (k < 0: half images of a |k| x |k| kernel)
                first += ((r[9] + 63) // 64) * ((r[10] + 63) // 64) * r[11] * r[11]
            else:
                first += ((r[9] + 31) // 32) * ((r[10] + 31) // 32)
        if len(_WIMG_TABLES) > 64:
            _WIMG_TABLES.clear()
        hit = _WIMG_TABLES[key] = (torch.from_numpy(arr.view(np.uint8).copy()).to(dev), first)
    return hit


def weight_images(records, dev):
    """The images of ``records`` = tuples (src_ptr, so, si, sy, sx, f_hi, f_lo, d_hi, d_lo, cout, cin, k): one launch of
    omnihd_weight_images_cl for the weights in channels_last memory (si == 1: the training step's), one of omnihd_weight_images
    for the others."""
    if not records:
        return
    cl = [r for r in records if r[2] == 1 and _env("OMNIHD_WEIGHT_IMAGES_CL", "1") != "0"]
    rest = [r for r in records if not (r[2] == 1 and _env("OMNIHD_WEIGHT_IMAGES_CL", "1") != "0")]
    with _on(dev):
        if cl:
            table, blocks = _weight_image_table(cl, dev, per_tap=True)
            check(lib().omnihd_weight_images_cl(table.data_ptr(), len(cl), blocks, _raw_stream()), "omnihd_weight_images_cl")
        if rest:
            table, blocks = _weight_image_table(rest, dev)
            check(lib().omnihd_weight_images(table.data_ptr(), len(rest), blocks, _raw_stream()), "omnihd_weight_images")


def refresh_split_shadows():
    """Bring every stale split image of an fp32 convolution weight (forward planes and, where a layer has asked for them, the
    data-gradient planes) up to date with ONE launch; a training loop calls it right after the optimiser step, like
    ``refresh_bf16_shadows``.  Returns how many layers were refreshed.  (Without it ``split_weight`` refreshes layer by layer.)"""
    per_dev = {}
    touched = []
    layers = {}
    for (wid, dgrad), (ref, ver, planes) in list(_SPLIT_SHADOW.items()):
        w = ref()
        if w is None:
            del _SPLIT_SHADOW[(wid, dgrad)]
            continue
        if ver == _wver(w) or planes[0].device != w.device or w.dtype != torch.float32:
            continue
        layers.setdefault(wid, [w, None, None])[2 if dgrad else 1] = planes
        touched.append((wid, dgrad, ref, w, planes))
    for wid, (w, fwd, dg) in layers.items():
        if fwd is None:                      # the data-gradient image alone is stale (cannot happen in a training loop): lazy path
            touched = [t for t in touched if t[0] != wid]
            continue
        cout, cin, k, _ = w.shape
        so, si, sy, sx = w.stride()
        rec = (w.data_ptr(), so, si, sy, sx, fwd[0].data_ptr(), fwd[1].data_ptr(), 0 if dg is None else dg[0].data_ptr(),
               0 if dg is None else dg[1].data_ptr(), cout, cin, k)
        per_dev.setdefault(w.device, []).append(rec)
    for dev, recs in per_dev.items():
        weight_images(recs, dev)
    for wid, dgrad, ref, w, planes in touched:
        _SPLIT_SHADOW[(wid, dgrad)] = (ref, _wver(w), planes)
    return len(layers)


def conv_fwd_split(xs, ws, bias=None, dilation=1, tile=0):
    """fp32-grade y = conv2d(x, w, bias, stride 1, padding = dilation*(k//2)) from split operands: xs = (x_hi, x_lo)
    (B,Cin,H,W) bf16 channels-last, ws = (w_hi, w_lo) (Cout,Cin,k,k) bf16 channels_last -> (B,Cout,H,W) fp32 channels-last."""
    for t in xs:
        _want_cl(t, "x plane")
    for t in ws:
        if t.dtype != torch.bfloat16 or t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
            raise TypeError("weight planes must be 4-D bf16 tensors in channels_last memory format")
    B, cin, H, W = xs[0].shape
    cout, k = ws[0].shape[0], ws[0].shape[2]
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=xs[0].device, memory_format=torch.channels_last)
    with _on(y.device):
        _conv_timed("split", (cin, cout, k, H, W), lambda: check(
            lib().omnihd_conv_fwd_split(xs[0].data_ptr(), xs[1].data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(),
                                        None if bias is None else _f32c(bias).data_ptr(), y.data_ptr(), B, H, W, cin, cout, k,
                                        int(dilation), int(tile), _raw_stream()), "omnihd_conv_fwd_split"))
    return y


def conv_split_geometry(x_shape, cout, k, stride, padding, dilation, groups=1):
    """(forward ok, data gradient ok, weight gradient ok) for the split kernels on a convolution geometry."""
    B, cin, H, W = x_shape
    s, p, d = _pair_same(stride), _pair_same(padding), _pair_same(dilation)
    if groups != 1 or s is None or p is None or d is None or k not in (1, 3):
        return False, False, False
    same = s == 1 and p == d * (k // 2)
    # "igemm": the stride-1 kernels of csrc/conv_igemm.hip (256-wide tiles, row-shift reuse); "gen": the general kernel of
    # csrc/conv_gen.hip (any stride / padding, channel counts that are multiples of 8) — both truthy
    fwd = "igemm" if (same and cin % 64 == 0 and cout % 8 == 0 and B * H * W < 2 ** 30) else False
    dgrad = "igemm" if (same and cout % 64 == 0 and cin % 8 == 0 and B * H * W < 2 ** 30) else False
    if not fwd and cout % 8 == 0 and conv_gen_supported(0, x_shape, cout, k, s, p, d):
        fwd = "gen"
    if not dgrad and cin % 8 == 0 and conv_gen_supported(1, x_shape, cout, k, s, p, d):
        dgrad = "gen"
    Ho, Wo = (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1
    wgrad = cin % 8 == 0 and cout % 8 == 0 and Ho > 0 and Wo > 0 and wgrad_nhwc_preferred(B, H, W, cin, Ho, Wo, cout, k, s, p, d)
    return fwd, dgrad, wgrad


# per-geometry measured choice between the split kernels and MIOpen's fp32 kernels (OMNIHD_FP32_CONV = tune | split | miopen)
_SPLIT_CHOICE = _ChoiceTable("split")


def _clock(fn, dev, n=3, warm=1):
    """Milliseconds of ``n`` back-to-back calls after ``warm`` untimed ones; OMNIHD_TUNE_REPEATS > 1 (used when the persisted
    choice table is captured) repeats the measurement and keeps the minimum."""
    best = None
    for _ in range(max(1, int(_env("OMNIHD_TUNE_REPEATS", "1")))):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        t = e0.elapsed_time(e1)
        best = t if best is None else min(best, t)
    return best


def _fp32_policy():
    return "split" if deterministic() else _env("OMNIHD_FP32_CONV", "tune")


def _split_pick(key, run_split, run_miopen, dev):
    policy = _fp32_policy()
    if policy in ("split", "f16"):          # (f16: layers the half form does not take run on the split kernels, never on a timing race)
        return run_split()
    if policy == "miopen":
        return run_miopen()
    choice = _SPLIT_CHOICE.get(key)
    if choice is None:
        choice = _SPLIT_CHOICE.measured(key, "split" if _clock(run_split, dev) <= _clock(run_miopen, dev) else "miopen")
    return run_split() if choice == "split" else run_miopen()


def split_choices():
    return dict(_SPLIT_CHOICE)


def conv_split_all_miopen(x_shape, cout, k, stride, padding, dilation, device_index):
    """True once every direction the split kernels could take for this geometry has been measured in MIOpen's favour: the
    layer is then a plain torch convolution again (no operand split, no Python in its backward)."""
    if _fp32_policy() != "tune":
        return False
    geo = (tuple(x_shape), cout, k, stride[0], padding[0], dilation[0], device_index)
    oks = conv_split_geometry(x_shape, cout, k, stride, padding, dilation)
    return all(_SPLIT_CHOICE.get((d,) + geo) == "miopen" for d, ok in zip(("fwd", "dgrad", "wgrad"), oks) if ok)


class _ConvSplit(torch.autograd.Function):
    """fp32 convolution of the reference-precision step on the split kernels: forward and data gradient on
    omnihd_conv_fwd_split, weight gradient on omnihd_conv_wgrad_split (one launch, hi*hi + hi*lo + lo*hi into fp32 tiles);
    per geometry and direction the measured faster of that and MIOpen's fp32 kernel runs.  The input is saved as its two
    bf16 planes (the same bytes as the fp32 tensor)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, grad_planes_only=False):
        dev = x.device
        ctx.grad_planes_only = bool(grad_planes_only)
        x = x.contiguous(memory_format=torch.channels_last)
        k = weight.shape[2]
        geo = (tuple(x.shape), weight.shape[0], k, stride[0], padding[0], dilation[0], dev.index)
        ok_f, ok_d, ok_w = conv_split_geometry(x.shape, weight.shape[0], k, stride, padding, dilation)
        # what the backward needs of x: its two bf16 planes for the split weight-gradient chain, or x itself where MIOpen's
        # fp32 weight gradient has measured faster for this geometry (no reconstruction of x from the planes then)
        wg_miopen = (not ok_w) or (_fp32_policy() == "tune" and _SPLIT_CHOICE.get(("wgrad",) + geo) == "miopen")
        use_split_fwd = ok_f and not (_fp32_policy() == "tune" and _SPLIT_CHOICE.get(("fwd",) + geo) == "miopen")
        xs = None
        if use_split_fwd or not wg_miopen:
            xs = take_planes(x)
            if xs is None:
                xs = split_f32(x)
        if wg_miopen:
            ctx.save_for_backward(x, weight)
        else:
            ctx.save_for_backward(xs[0], xs[1], weight)
        ctx.x_is_full = wg_miopen
        ctx.conv = (list(stride), list(padding), list(dilation), geo, ok_d, ok_w)
        ctx.has_bias = bias is not None
        ctx.param_dtypes = (weight.dtype, None if bias is None else bias.dtype)
        run_miopen = lambda: torch.nn.functional.conv2d(x, weight.detach(), None if bias is None else bias.detach(), stride,
                                                         padding, dilation)
        if not use_split_fwd:
            return run_miopen()
        if ok_f == "gen":
            run_split = lambda: conv_gen(0, xs, split_weight(weight), None if bias is None else bias.detach(), tuple(x.shape),
                                         weight.shape[0], k, stride[0], padding[0], dilation[0])
        else:
            run_split = lambda: conv_fwd_split(xs, split_weight(weight), None if bias is None else bias.detach(), dilation[0])
        return _split_pick(("fwd",) + geo, run_split, run_miopen, dev)

    @staticmethod
    def backward(ctx, g):
        if ctx.x_is_full:
            x_saved, weight = ctx.saved_tensors
            x_hi = x_lo = None
            x_shape = x_saved.shape
        else:
            x_hi, x_lo, weight = ctx.saved_tensors
            x_saved, x_shape = None, x_hi.shape
        stride, padding, dilation, geo, ok_d, ok_w = ctx.conv
        dev = g.device
        g_in = g
        if getattr(g_in, "_omnihd_planes_only", False) and not ctx.grad_planes_only:
            raise RuntimeError("a planes-only gradient reached a convolution that did not ask for one (conv_bn_act's contract)")
        g = g.float().contiguous(memory_format=torch.channels_last)
        want_w_split = ok_w and ctx.needs_input_grad[1] and not ctx.x_is_full
        gs = None
        if (ok_d and ctx.needs_input_grad[0]) or want_w_split:
            gs = take_planes(g_in) if g is g_in else None
            if gs is None:
                if ctx.grad_planes_only:
                    raise RuntimeError("the BatchNorm behind this convolution promised its input gradient as planes and did not "
                                       "deliver them (conv_bn_act's contract)")
                gs = split_f32(g)
        gx = gw = gb = None
        x_f32 = []

        def x_full():                      # only for MIOpen's weight gradient: hi + lo reproduces x to 2^-17
            if x_saved is not None:
                return x_saved
            if not x_f32:
                x_f32.append(x_hi.float().add_(x_lo))
            return x_f32[0]

        def weight_gradient(out=None):
            run_miopen = lambda: torch.ops.aten.convolution_backward(g, x_full(), weight.detach(), None, stride, padding, dilation,
                                                                     False, [0, 0], 1, [False, True, False])[1]
            if want_w_split:
                k = weight.shape[2]
                run_split = lambda: conv_wgrad_split((x_hi, x_lo), gs, k, stride[0], padding[0], dilation[0], out=out)
                return _split_pick(("wgrad",) + geo, run_split, run_miopen, dev).to(ctx.param_dtypes[0])
            return run_miopen().to(ctx.param_dtypes[0])

        def into_view(view, gw):
            """The weight gradient inside the reducer's bucket: our kernel wrote it there already, a library result is copied."""
            if gw.data_ptr() != view.data_ptr():
                view.copy_(gw)
            _DDP["direct"] += 1
            return view.detach()             # a fresh alias in the parameter's layout: autograd keeps it as .grad without a kernel

        # Weight gradient beside the data gradient (OMNIHD_WGRAD_OVERLAP, one rank): nothing reads a weight gradient before the end
        # of the backward pass, so its kernels go to a side stream that the autograd engine's final callback joins
        # (wgrad_overlap_join); the data-gradient chain on the main stream no longer waits for them, and the tails of either
        # fill the other's idle CUs.  By default only for the layers BEHIND the pooling backward (wgrad_overlap_arm).
        side = _wgrad_side_stream(dev, weight) if ctx.needs_input_grad[1] else None
        if side is None and ctx.needs_input_grad[1]:
            FAST_PATHS["wgrad_in_line"] += 1
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(dev))
            for tns in (g, g_in, x_hi, x_lo, x_saved) + (tuple(gs) if gs is not None else ()):
                if tns is not None:
                    tns.record_stream(side)              # allocated on the main stream, read on the side stream
            view = _ddp_bucket_view(weight)
            if view is not None:
                _VIEW_WRITTEN.add(id(weight))        # a second convolution on this weight must not write the view again
            with torch.cuda.stream(side):
                gw = weight_gradient(view if _view_writable(view, weight) else None)
                if view is not None:
                    # under DistributedDataParallel: straight into the reducer's bucket.  What autograd gets back is a fresh
                    # alias of that memory in the parameter's layout — AccumulateGrad keeps it as .grad without a kernel, the
                    # reducer sees "already in the bucket" and copies nothing, and the bucket's all-reduce (our comm hook)
                    # waits for this stream.  No kernel of the caller's stream touches the gradient before the pass ends.
                    gw = into_view(view, gw)
                elif gw.stride() != weight.stride():
                    # autograd keeps a gradient that has the parameter's layout as it is; any other one it would COPY on the
                    # main stream, before this stream is done
                    gw = torch.empty_like(weight).copy_(gw)
            # allocated on the side stream, consumed (clip, AdamW, zero_grad's free) on the caller's
            if view is None:
                gw.record_stream(torch.cuda.current_stream(dev))
        if ctx.needs_input_grad[0]:
            # (an uninitialised fp32 stand-in for the input: only its shape / layout matter to the data gradient)
            x_like = lambda: torch.empty(x_shape, dtype=torch.float32, device=dev, memory_format=torch.channels_last)
            run_miopen = lambda: torch.ops.aten.convolution_backward(g, x_like(), weight.detach(), None, stride, padding, dilation,
                                                                     False, [0, 0], 1, [True, False, False])[0]
            if ok_d == "gen":
                run_split = lambda: conv_gen(1, gs, split_weight(weight, dgrad=True), None, tuple(x_shape), weight.shape[0],
                                             weight.shape[2], stride[0], padding[0], dilation[0])
                gx = _split_pick(("dgrad",) + geo, run_split, run_miopen, dev)
            elif ok_d:
                run_split = lambda: conv_fwd_split(gs, split_weight(weight, dgrad=True), None, dilation[0])
                gx = _split_pick(("dgrad",) + geo, run_split, run_miopen, dev)
            else:
                gx = run_miopen()
        elif ok_d:
            _SPLIT_CHOICE.setdefault(("dgrad",) + geo, "miopen")          # never asked for: nothing to measure
        if ctx.needs_input_grad[1] and side is None:
            # in line; under a hooked reducer still straight into the bucket view (the reducer then has nothing to copy)
            view = None
            if weight.is_leaf and weight.grad is None and not torch.is_grad_enabled() and _WGRAD_ENGINE_OK and want_w_split:
                _wgrad_pass_begin()                  # (the per-pass sets below belong to THIS backward pass)
                # A weight that feeds several convolutions of one pass: ``weight.grad`` stays None until autograd
                # has summed ALL its gradients, so the first use's alias of the bucket view is still pending when the second use
                # arrives.  Writing the view again would overwrite the first gradient (autograd would then sum two aliases of
                # one buffer: 2*g2 instead of g1 + g2) — from the second sighting on the gradient goes into a fresh tensor and
                # autograd sums, the reducer copies.
                if id(weight) not in _VIEW_WRITTEN:
                    view = _ddp_bucket_view(weight)
            if _view_writable(view, weight):
                _VIEW_WRITTEN.add(id(weight))
                gw = into_view(view, weight_gradient(view))
            else:
                gw = weight_gradient()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            n, c, h, w = g.shape
            gb = column_sums(g.permute(0, 2, 3, 1).reshape(n * h * w, c)).to(ctx.param_dtypes[1])
        return gx, gw, gb, None, None, None, None


# counters of the optional fast paths actually taken in this process (bench.py: `fast_paths`)
FAST_PATHS = {"wgrad_side_stream": 0, "wgrad_in_line": 0, "dual_stream_forward": 0, "single_stream_forward": 0}


def fast_paths_report():
    """What ran, from live counters — not from the environment switches: the kept pooling buffers and the weight-gradient side
    stream rest on private torch hooks (``torch._C._storage_Use_Count``, ``torch._C._current_graph_task_id`` +
    ``queue_callback``) that are probed and fall back silently when a torch build lacks them."""
    from . import plan as _plan
    calls = max(1, _plan.FAST_PATHS["pool_fwd_calls"])
    wg = FAST_PATHS["wgrad_side_stream"] + FAST_PATHS["wgrad_in_line"]
    fw = FAST_PATHS["dual_stream_forward"] + FAST_PATHS["single_stream_forward"]
    return {"kept_output": {"active": _plan.FAST_PATHS["kept_output"] > 0, "share_of_pool_forwards": round(_plan.FAST_PATHS["kept_output"] / calls, 3),
                            "private_hook_ok": bool(_plan._use_count_works())},
            "direct_fwd": {"active": _plan.FAST_PATHS["direct_fwd"] > 0, "share_of_pool_forwards": round(_plan.FAST_PATHS["direct_fwd"] / calls, 3)},
            "wgrad_overlap": {"active": FAST_PATHS["wgrad_side_stream"] > 0, "share_of_split_weight_gradients": round(FAST_PATHS["wgrad_side_stream"] / max(1, wg), 3),
                              "private_hooks_ok": bool(_WGRAD_ENGINE_OK), "ddp": ddp_overlap_info()},
            "dual_stream": {"active": FAST_PATHS["dual_stream_forward"] > 0, "share_of_forwards": round(FAST_PATHS["dual_stream_forward"] / max(1, fw), 3)},
            "device_plans_built": int(__import__("omnihd_amd.pool_plan", fromlist=["BUILDS"]).BUILDS["device_plans"]),
            "choice_table_misses": int(_CHOICE_INFO["misses"]),
            # uploads of a weight-image table (a BLOCKING host-to-device copy each): 0 in a steady step
            "weight_table_uploads": int(WIMG_STATS["miss"])}


def fast_paths_reset():
    from . import plan as _plan
    for d in (FAST_PATHS, _plan.FAST_PATHS, WIMG_STATS):
        for k in d:
            d[k] = 0


_WGRAD_SIDE = {}
_WGRAD_SIDE_USED = set()
_WGRAD_SEEN = set()         # ids of the weights whose gradient went to the side stream in this backward pass
_VIEW_WRITTEN = set()       # ids of the weights whose gradient was written into the reducer's bucket view in this backward pass
_WGRAD_ARMED = []           # non-empty: the pooling backward of this backward pass has been launched (see wgrad_overlap_arm)
_WGRAD_PASS = [None]        # autograd graph-task id of the backward pass the two above belong to
# the two private hooks of the autograd engine this rests on; a torch without them keeps the in-line path
_WGRAD_ENGINE_OK = hasattr(torch._C, "_current_graph_task_id") and hasattr(torch.autograd.Variable._execution_engine, "queue_callback")


# ---- weight-gradient overlap under DistributedDataParallel (round 5) ---------------------------------------------------------
# The reference overlaps its reducer with backward on every rank (bevformer/apis/mmdet_train.py:76-80); round 4 switched the side
# stream OFF whenever a process group existed, so the N = 1 headline rested on an optimisation N > 1 could not use.  Now the N > 1
# step is the N = 1 step: `ddp_wgrad_overlap(ddp)` registers a communication hook on the reducer that (a) all-reduces a bucket on a
# communication stream that waits for BOTH the caller's stream and the weight-gradient side stream, and (b) remembers, per
# parameter, the reducer's view of its gradient inside the bucket (gradient_as_bucket_view=True) once the reducer has re-bucketed
# (it does so once, before the second forward): the side stream then writes weight gradients straight into those views.
_DDP = {"ref": None, "views": {}, "settled": False, "comm": {}, "hook_calls": 0, "direct": 0, "layout": {}, "dirty": False}


def ddp_wgrad_overlap(ddp):
    """Register the bucket hook on a DistributedDataParallel module (built with gradient_as_bucket_view=True).  Returns True when
    registered.  Without it a process group keeps every weight gradient in line, as before."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    if not isinstance(ddp, DDP) or not getattr(ddp, "gradient_as_bucket_view", False):
        return False
    _DDP.update(ref=weakref.ref(ddp), views={}, settled=False, hook_calls=0, direct=0, layout={}, dirty=False)
    ddp.register_comm_hook(None, _ddp_bucket_hook)
    return True


def _dense(t):
    try:
        from torch._prims_common import is_non_overlapping_and_dense
        return bool(is_non_overlapping_and_dense(t))
    except Exception:
        return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def _ddp_bucket_hook(_state, bucket):
    import torch.distributed as dist
    ddp = _DDP["ref"]() if _DDP["ref"] is not None else None
    buf = bucket.buffer()
    group = ddp.process_group if ddp is not None else None
    world = dist.get_world_size(group)
    _DDP["hook_calls"] += 1
    if ddp is not None:
        # Are the reducer's buckets settled?  It re-buckets once, before its second forward, in the order the gradients arrived
        # (new buffers, new views).  The reducer calls this hook under its own mutex, so its logging data cannot be asked here
        # (that deadlocks); instead: a pass whose every bucket (index, buffer address, size) is what the previous pass saw.
        # Any change drops the remembered views and starts over.
        key = (int(buf.data_ptr()), int(buf.numel()))
        changed = _DDP["layout"].get(bucket.index()) != key
        if changed:
            _DDP["layout"][bucket.index()] = key
            _DDP["dirty"] = True
            _DDP["settled"] = False
            _DDP["views"] = {}
        # the reducer's own views follow the parameter's strides (dense parameters); GradBucket.gradients() hands out row-major
        # views of the same memory, so only their offsets are taken from it.  Once the buckets are settled (same buffers pass after
        # pass) the views are known: the walk over the bucket's parameters — ≈60 tensor views built per step in 26 calls, host time
        # inside backward that a slow host does not hide — is skipped.
        if changed or not _DDP["settled"]:
            for p, g in zip(bucket.parameters(), bucket.gradients()):
                hit = _DDP["views"].get(id(p))
                if hit is None or hit[1].data_ptr() != g.data_ptr():
                    v = buf.as_strided(p.size(), p.stride(), g.storage_offset()) if _dense(p) else g
                    _DDP["views"][id(p)] = (weakref.ref(p), v)
        if bucket.is_last():
            if not _DDP["dirty"]:
                _DDP["settled"] = True
            _DDP["dirty"] = False
    if buf.is_cuda:
        dev = buf.device
        comm = _DDP["comm"].get(dev.index)
        if comm is None:
            comm = _DDP["comm"][dev.index] = torch.cuda.Stream(device=dev)
        comm.wait_stream(torch.cuda.current_stream(dev))
        if dev.index in _WGRAD_SIDE:
            comm.wait_stream(_WGRAD_SIDE[dev.index])
        with torch.cuda.stream(comm):
            if world > 1:
                buf.div_(world)
            fut = dist.all_reduce(buf, group=group, async_op=True).get_future()
    else:
        if world > 1:
            buf.div_(world)
        fut = dist.all_reduce(buf, group=group, async_op=True).get_future()
    return fut.then(lambda f: f.value()[0])


def _ddp_bucket_view(weight):
    """The reducer's view of ``weight``'s gradient inside its bucket, or None (no hooked reducer, buckets not settled yet, the
    reducer not synchronising this pass — DDP.no_sync() — or a stale entry)."""
    if _DDP["ref"] is None or not _DDP["settled"]:
        return None
    ddp = _DDP["ref"]()
    if ddp is None or not ddp.require_backward_grad_sync:
        return None
    hit = _DDP["views"].get(id(weight))
    if hit is None or hit[0]() is not weight:
        return None
    v = hit[1]
    if v.shape != weight.shape or v.stride() != weight.stride() or v.dtype != weight.dtype or v.device != weight.device:
        return None
    return v


def _view_writable(view, weight):
    """Can our weight-gradient kernel write straight into ``view``?  fp32, 4-D, (Cout,k,k,Cin) memory."""
    return (view is not None and view.dtype == torch.float32 and view.dim() == 4 and view.permute(0, 2, 3, 1).is_contiguous())


def ddp_overlap_info():
    """{'hooked', 'settled', 'views', 'hook_calls'} — what bench.py reports as fast_paths.wgrad_overlap under a process group."""
    return {"hooked": _DDP["ref"] is not None and _DDP["ref"]() is not None, "settled": bool(_DDP["settled"]),
            "views": len(_DDP["views"]), "hook_calls": int(_DDP["hook_calls"]), "direct_writes": int(_DDP["direct"])}


def _wgrad_side_stream(dev, weight):
    """The side stream for the weight gradient of ``weight``, or None: OMNIHD_WGRAD_OVERLAP=0; a process group exists (a DDP
    reducer, also a one-rank one, copies gradients into its buckets as autograd accumulates them, on its own stream); the
    parameter already holds a gradient or carries hooks (autograd would then run kernels on the gradient on the main stream,
    before the side stream is done); the backward pass builds a graph; or — the default mode — the pooling backward of this pass
    has not run yet (OMNIHD_WGRAD_OVERLAP=all: every layer from the start of the pass).  The first use inside a backward pass queues
    ``wgrad_overlap_join`` as a final callback of the autograd engine, so whoever called ``backward`` finds the gradients
    complete on its stream — no caller has to know."""
    mode = _env("OMNIHD_WGRAD_OVERLAP", "1")
    if mode == "0" or torch.is_grad_enabled() or not _WGRAD_ENGINE_OK:
        return None
    if not weight.is_leaf or weight.grad is not None or weight._backward_hooks or getattr(weight, "_post_accumulate_grad_hooks", None):
        return None
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and _ddp_bucket_view(weight) is None:
        # a process group without our comm hook on the reducer (or before the reducer's buckets have settled): a DDP reducer
        # copies gradients into its buckets as autograd accumulates them, on the caller's stream
        return None
    _wgrad_pass_begin()
    if mode != "all" and not _WGRAD_ARMED:
        return None
    # A weight that feeds SEVERAL convolutions of one pass: the engine sums their gradients on the caller's stream as soon as the
    # last one has arrived — from the second sighting on, the caller's stream first waits for what the side stream holds and the
    # layer stays in line (tests/test_conv_split_gpu.py::test_shared_weight...)
    if id(weight) in _WGRAD_SEEN or id(weight) in _VIEW_WRITTEN:
        if dev.index in _WGRAD_SIDE_USED:
            torch.cuda.current_stream(dev).wait_stream(_WGRAD_SIDE[dev.index])
        return None
    _WGRAD_SEEN.add(id(weight))
    s = _WGRAD_SIDE.get(dev.index)
    if s is None:
        # (stream priorities do not help here: this device offers two, high and normal, so the side stream cannot be put BELOW the
        # default stream; the whole step on a high-priority stream instead measured 54 ms, not 48.5)
        s = _WGRAD_SIDE[dev.index] = torch.cuda.Stream(device=dev)
    _WGRAD_SIDE_USED.add(dev.index)
    FAST_PATHS["wgrad_side_stream"] += 1
    return s


def _wgrad_pass_begin():
    """First touch of the overlap state inside a backward pass (autograd's graph-task id tells passes apart): this pass's join
    is queued as a final callback of the engine; what an aborted pass left behind is joined first."""
    task = torch._C._current_graph_task_id()
    if _WGRAD_PASS[0] != task:
        if _WGRAD_SIDE_USED:
            wgrad_overlap_join()
        _WGRAD_ARMED.clear()
        _WGRAD_SEEN.clear()
        _VIEW_WRITTEN.clear()
        _WGRAD_PASS[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(wgrad_overlap_join)


def wgrad_overlap_join():
    """End of a backward pass: the current stream waits for the weight gradients that were computed on the side stream."""
    for idx in list(_WGRAD_SIDE_USED):
        torch.cuda.current_stream(idx).wait_stream(_WGRAD_SIDE[idx])
    _WGRAD_SIDE_USED.clear()
    _WGRAD_ARMED.clear()
    _WGRAD_SEEN.clear()
    _VIEW_WRITTEN.clear()
    _WGRAD_PASS[0] = None


def wgrad_overlap_arm():
    """Called by the pooling backward once its kernel is enqueued: from here to the end of the backward pass (DepthNet and the
    image backbone: ~90 convolutions of small and middle size) the weight gradients go to the side stream.  The layers in front
    of it (heads, fusion, BEV encoder: few, GPU-filling kernels) keep theirs in line, so the bandwidth-bound pooling backward
    never shares the memory system with a matrix kernel of the side stream (119 us instead of 53 us in the step when it does,
    OMNIHD_WGRAD_OVERLAP=all with OMNIHD_POOL_BWD_EXCLUSIVE=0) and never waits for one.  Measured alternatives, same box: all layers
    + a fence in front of the pooling backward 48.46 ms, recording the front layers' work and enqueueing it behind the pooling
    kernel 48.23 ms (but that kernel then 64 us), this 48.47 ms, no overlap 50.0 ms."""
    if not _WGRAD_ENGINE_OK or torch._C._current_graph_task_id() < 0 or _env("OMNIHD_WGRAD_OVERLAP", "1") == "0":
        return
    _wgrad_pass_begin()
    if not _WGRAD_ARMED:
        _WGRAD_ARMED.append(True)


def wgrad_overlap_fence(dev):
    """Inside a backward pass: the current stream waits for the weight gradients enqueued so far (only OMNIHD_WGRAD_OVERLAP=all
    enqueues any in front of the pooling backward, which calls this)."""
    if dev.index in _WGRAD_SIDE_USED and _env("OMNIHD_POOL_BWD_EXCLUSIVE", "1") != "0":
        torch.cuda.current_stream(dev).wait_stream(_WGRAD_SIDE[dev.index])


def conv_split_supported(x, weight, stride, padding, dilation, groups=1):
    """fp32 device activations and at least one direction the split kernels take."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight.dtype == torch.float32 and weight.dim() == 4
            and weight.shape[2] == weight.shape[3] and weight.shape[1] == x.shape[1]):
        return False
    return any(conv_split_geometry(x.shape, weight.shape[0], weight.shape[2], stride, padding, dilation, groups))


def conv_split(x, weight, bias, stride, padding, dilation=(1, 1), grad_planes_only=False):
    if _fp32_policy() == "f16" and conv_f16_applies(x.shape, weight, stride, padding, dilation):
        return _ConvF16.apply(x, weight, bias, tuple(stride), tuple(padding), tuple(dilation))
    return _ConvSplit.apply(x, weight, bias, tuple(stride), tuple(padding), tuple(dilation), bool(grad_planes_only))


# ---------------------------------------------------------------------------------------------
# TF32-grade form of the fp32 step's convolutions (round 6; OMNIHD_FP32_CONV=f16): ONE half MFMA product per fp32 product
# ---------------------------------------------------------------------------------------------
# The reference trains with TF32 left on (tools/train.py:150-153): 11 significant bits per operand.  An IEEE half has the same 11
# bits; activations and weights are converted as they are, gradients with an exact power-of-two scale found per tensor (amax pass)
# whose inverse the consuming kernel applies.  Layers the half kernels do not take (strided, transposed, narrow) stay on the
# fp32-grade split kernels, so every layer of the step is at least TF32-grade.  Parity: tests/test_conv_f16_gpu.py.
_F16_SHADOW = {}


def cast_f16(t, scaled=False):
    """fp32 tensor (dense in its memory format) -> (half tensor of the same shape and strides, inverse scale).  ``scaled``: the
    values are multiplied by the power of two that brings the largest magnitude just below 2^15; the second result is a device
    scalar holding the inverse (what the kernels take as ``alpha``); without ``scaled`` it is None."""
    if not (t.is_cuda and t.dtype == torch.float32):
        raise TypeError("cast_f16 takes an fp32 CUDA(HIP) tensor")
    if not (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))):
        t = t.contiguous()
    out = torch.empty_strided(t.shape, t.stride(), dtype=torch.float16, device=t.device)
    mode, scratch = 0, None
    if scaled == "ring":
        mode, scratch = 2, _amax_slot(t.device)
    elif scaled:
        mode, scratch = 1, torch.empty(2, dtype=torch.float32, device=t.device)
    with _on(t.device):
        check(lib().omnihd_cast_f16(t.data_ptr(), t.numel(), mode, out.data_ptr(), _ptr(scratch), _raw_stream()), "omnihd_cast_f16")
    return out, (scratch[1:2] if scaled else None)


_AMAX_RING = {}
_AMAX_SLOTS = 2048


def _amax_slot(dev):
    """Two zeroed device words for a scaled cast whose scale is consumed by launches enqueued right behind it on the SAME stream
    (the backward of _ConvF16, or the backward of the BatchNorm / affine layer behind it, which accumulates the amax there):
    slots of a ring that a fill re-zeroes every _AMAX_SLOTS / 2 casts instead of a memset node per cast (92 fills per step in the
    first profile of the form).  Stream order makes the re-zeroing safe: it is enqueued behind every consumer of the slots it clears."""
    key = (dev.index, _raw_stream())
    e = _AMAX_RING.get(key)
    if e is None:
        e = _AMAX_RING[key] = [torch.zeros(2 * _AMAX_SLOTS, dtype=torch.float32, device=dev), 0, False]
    i = e[1]
    if i == _AMAX_SLOTS:
        i, e[2] = 0, True
    # the ring is re-zeroed HALF by half, each half when the index enters it: a slot handed out just before (a producer's backward
    # has accumulated its amax there, the consumer's cast is not enqueued yet) lies in the other half and stays intact
    if e[2] and (i == 0 or i == _AMAX_SLOTS // 2):
        e[0][2 * i:2 * i + _AMAX_SLOTS].zero_()
    e[1] = i + 1
    return e[0][2 * i:2 * i + 2]


def f16_weight(weight, dgrad=False):
    """Half image of an fp32 convolution weight in (Cout,k,k,Cin) memory — ``dgrad``: (Cin,k,k,Cout) with mirrored taps — cached
    while the parameter's version is unchanged (``refresh_f16_shadows`` rebuilds all stale ones with one launch)."""
    if weight.grad_fn is not None:
        # a temporary computed from a parameter (the block-diagonal matrix DCN rebuilds every forward): never seen again under this
        # id — converted directly, no cache entry and no table upload (a table miss is a blocking host-to-device copy)
        w = weight.detach().float()
        w = w.flip(2, 3).transpose(0, 1) if dgrad else w
        return cast_f16(w.contiguous(memory_format=torch.channels_last))[0]
    key = (id(weight), dgrad)
    e = _F16_SHADOW.get(key)
    if e is not None and e[0]() is weight and e[1] == _wver(weight) and e[2].device == weight.device:
        return e[2]
    cout, cin, k, _ = weight.shape
    shape = (cin, cout, k, k) if dgrad else (cout, cin, k, k)
    img = e[2] if (e is not None and e[0]() is weight and tuple(e[2].shape) == shape and e[2].device == weight.device) else \
        torch.empty(shape, dtype=torch.float16, device=weight.device, memory_format=torch.channels_last)
    w = weight.detach()
    so, si, sy, sx = w.stride()
    rec = (w.data_ptr(), so, si, sy, sx, 0 if dgrad else img.data_ptr(), 0, img.data_ptr() if dgrad else 0, 0, cout, cin, -k)
    if dgrad:
        # the kernel always writes the forward image too: give it the forward shadow (built here if need be)
        fwd = f16_weight(weight)
        rec = rec[:5] + (fwd.data_ptr(),) + rec[6:]
    weight_images([rec], weight.device)
    if len(_F16_SHADOW) > 4096:
        for k_ in [k_ for k_, v in _F16_SHADOW.items() if v[0]() is None]:
            del _F16_SHADOW[k_]
    _F16_SHADOW[key] = (weakref.ref(weight), _wver(weight), img)
    return img


def refresh_f16_shadows():
    """Bring every stale half image up to date with ONE launch (a training loop calls it behind the optimiser step)."""
    per_dev, touched, layers = {}, [], {}
    for (wid, dgrad), (ref, ver, img) in list(_F16_SHADOW.items()):
        w = ref()
        if w is None:
            del _F16_SHADOW[(wid, dgrad)]
            continue
        if ver == _wver(w) or img.device != w.device or w.dtype != torch.float32:
            continue
        layers.setdefault(wid, [w, None, None])[2 if dgrad else 1] = img
        touched.append((wid, dgrad, ref, w, img))
    for wid, (w, fwd, dg) in layers.items():
        if fwd is None:
            touched = [t for t in touched if t[0] != wid]
            continue
        cout, cin, k, _ = w.shape
        so, si, sy, sx = w.stride()
        per_dev.setdefault(w.device, []).append((w.data_ptr(), so, si, sy, sx, fwd.data_ptr(), 0, 0 if dg is None else dg.data_ptr(), 0,
                                                  cout, cin, -k))
    for dev, recs in per_dev.items():
        weight_images(recs, dev)
    for wid, dgrad, ref, w, img in touched:
        _F16_SHADOW[(wid, dgrad)] = (ref, _wver(w), img)
    return len(layers)


def conv_fwd_f16(x16, w16, bias=None, alpha=None, dilation=1, tile=0):
    """y = alpha * conv2d(x16, w16, stride 1, padding = dilation*(k//2)) + bias on half operands (channels_last) -> fp32 channels_last."""
    for t in (x16, w16):
        if t.dtype != torch.float16 or t.dim() != 4 or not t.is_contiguous(memory_format=torch.channels_last):
            raise TypeError("conv_fwd_f16 takes 4-D half tensors in channels_last memory format")
    B, cin, H, W = x16.shape
    cout, k = w16.shape[0], w16.shape[2]
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x16.device, memory_format=torch.channels_last)
    with _on(y.device):
        _conv_timed("f16", (cin, cout, k, H, W), lambda: check(
            lib().omnihd_conv_fwd_f16(x16.data_ptr(), w16.data_ptr(), None if bias is None else _f32c(bias).data_ptr(), y.data_ptr(),
                                      _ptr(alpha), B, H, W, cin, cout, k, int(dilation), int(tile), _raw_stream()), "omnihd_conv_fwd_f16"))
    return y


def conv_wgrad_f16(x16, g16, alpha, kernel_size, stride=1, padding=0, dilation=1):
    """dW (Cout,Cin,k,k) fp32 (channels_last memory) = alpha * weight gradient from half operands (omnihd_conv_wgrad_nhwc_f16)."""
    B, cin, H, W = x16.shape
    _, cout, Ho, Wo = g16.shape
    k = int(kernel_size)
    dev = x16.device
    g11 = (B, H, W, cin, Ho, Wo, cout, k, int(stride), int(padding), int(dilation))
    if not wgrad_nhwc_preferred(*g11):
        raise ValueError(f"conv_wgrad_f16: the NHWC weight-gradient kernel does not take geometry {g11}")
    dw = torch.empty((cout, k, k, cin), dtype=torch.float32, device=dev)
    L = lib()
    with _on(dev):
        nbytes = _SIZE_CACHE.get(("nhwc",) + g11)
        if nbytes is None:
            nbytes = _SIZE_CACHE[("nhwc",) + g11] = L.omnihd_conv_wgrad_nhwc_workspace_bytes(*g11)
        ws = _wgrad_workspace(nbytes, dev)
        check(L.omnihd_conv_wgrad_nhwc_f16(x16.data_ptr(), g16.data_ptr(), dw.data_ptr(), _ptr(alpha), *g11, ws.data_ptr(), ws.numel(),
                                           _raw_stream()), "omnihd_conv_wgrad_nhwc_f16")
    return dw.permute(0, 3, 1, 2)


def conv_f16_applies(x_shape, weight, stride, padding, dilation):
    """The half kernels take the layer in all three directions: stride-1 'same' 1x1 / 3x3, Cin and Cout multiples of 64."""
    if weight.dim() != 4 or weight.shape[2] != weight.shape[3] or weight.dtype != torch.float32:
        return False
    ok_f, ok_d, ok_w = conv_split_geometry(x_shape, weight.shape[0], weight.shape[2], stride, padding, dilation)
    return ok_f == "igemm" and ok_d == "igemm" and bool(ok_w)


_CL = torch.channels_last


def _f16_plane(t, mode, scratch, L, st):
    """cast_f16 without its argument checks (the caller holds a dense fp32 device tensor): one allocation, one library call."""
    out = torch.empty_like(t, dtype=torch.float16)              # preserve_format: a dense tensor keeps its strides
    check(L.omnihd_cast_f16(t.data_ptr(), t.numel(), mode, out.data_ptr(), None if scratch is None else scratch.data_ptr(), st),
          "omnihd_cast_f16")
    return out


class _ConvF16(torch.autograd.Function):
    """fp32 convolution in the TF32-grade form: forward, data gradient (omnihd_conv_fwd_f16) and weight gradient
    (omnihd_conv_wgrad_nhwc_f16) on half operands with fp32 accumulation; the input is saved as its half plane.
    (The step is host-bound in this form — scripts/lab/host_profile.py — so the calls below go to the library directly: the
    checked wrappers ``cast_f16`` / ``conv_fwd_f16`` / ``conv_wgrad_f16`` are the public faces of the same entry points.)"""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation):
        if not x.is_contiguous(memory_format=_CL):
            x = x.contiguous(memory_format=_CL)              # (a new tensor: no producer's tag on it)
        dev = x.device
        L = lib()
        B, cin, H, W = x.shape
        cout, _, k, _ = weight.shape
        w16 = f16_weight(weight)
        y = torch.empty((B, cout, H, W), dtype=torch.float32, device=dev, memory_format=_CL)
        with _on(dev):
            st = _raw_stream()
            x16 = take_half(x)                               # written by x's producer (BatchNorm / affine epilogue), or cast here
            if x16 is None:
                x16 = _f16_plane(x, 0, None, L, st)
            _conv_timed("f16", (cin, cout, k, H, W), lambda: check(
                L.omnihd_conv_fwd_f16(x16.data_ptr(), w16.data_ptr(), None if bias is None else _f32c(bias.detach()).data_ptr(),
                                      y.data_ptr(), None, B, H, W, cin, cout, k, dilation[0], 0, st), "omnihd_conv_fwd_f16"))
        ctx.save_for_backward(x16, weight)
        ctx.conv = (stride[0], padding[0], dilation[0])
        ctx.has_bias = bias is not None
        ctx.param_dtypes = (weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        x16, weight = ctx.saved_tensors
        s, p, d = ctx.conv
        amax = getattr(g, "_omnihd_amax", None)              # (slot, version): g's producer has accumulated max |g| on the device
        if g.dtype != torch.float32:
            g, amax = g.float(), None
        if not g.is_contiguous(memory_format=_CL):
            g, amax = g.contiguous(memory_format=_CL), None
        if amax is not None and amax[1] != g._version:
            amax = None
        gx = gw = gb = None
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_x or need_w:
            dev = g.device
            L = lib()
            B, cin, H, W = x16.shape
            cout, _, k, _ = weight.shape
            wd = f16_weight(weight, dgrad=True) if need_x else None
            with _on(dev):
                st = _raw_stream()
                slot = amax[0] if amax is not None else _amax_slot(dev)
                g16 = _f16_plane(g, 3 if amax is not None else 2, slot, L, st)
                FAST_PATHS["f16_amax_from_producer" if amax is not None else "f16_amax_pass"] = \
                    FAST_PATHS.get("f16_amax_from_producer" if amax is not None else "f16_amax_pass", 0) + 1
                inv = slot.data_ptr() + 4
                if need_x:
                    gx = torch.empty((B, cin, H, W), dtype=torch.float32, device=dev, memory_format=_CL)
                    _conv_timed("f16", (cout, cin, k, H, W), lambda: check(
                        L.omnihd_conv_fwd_f16(g16.data_ptr(), wd.data_ptr(), None, gx.data_ptr(), inv, B, H, W, cout, cin, k, d, 0, st),
                        "omnihd_conv_fwd_f16"))
                if need_w:
                    FAST_PATHS["wgrad_in_line"] += 1
                    g11 = (B, H, W, cin, H, W, cout, k, s, p, d)
                    nbytes = _SIZE_CACHE.get(("nhwc",) + g11)
                    if nbytes is None:
                        nbytes = _SIZE_CACHE[("nhwc",) + g11] = L.omnihd_conv_wgrad_nhwc_workspace_bytes(*g11)
                    ws = _wgrad_workspace(nbytes, dev)
                    dw = torch.empty((cout, k, k, cin), dtype=torch.float32, device=dev)
                    check(L.omnihd_conv_wgrad_nhwc_f16(x16.data_ptr(), g16.data_ptr(), dw.data_ptr(), inv, *g11, ws.data_ptr(), ws.numel(), st),
                          "omnihd_conv_wgrad_nhwc_f16")
                    gw = dw.permute(0, 3, 1, 2)
                    if ctx.param_dtypes[0] != torch.float32:
                        gw = gw.to(ctx.param_dtypes[0])
        if ctx.has_bias and ctx.needs_input_grad[2]:
            n, c, h, w = g.shape
            gb = column_sums(g.permute(0, 2, 3, 1).reshape(n * h * w, c)).to(ctx.param_dtypes[1])
        return gx, gw, gb, None, None, None


def conv_grad_planes_ok(x_shape, weight, bias, stride, padding, dilation, device_index):
    """May the backward of this fp32 convolution take its output gradient as hi / lo planes ONLY?  Yes when every consumer of
    that gradient inside ``_ConvSplit.backward`` is a split kernel: no bias (its gradient sums the fp32 tensor), data and weight
    gradient on the split kernels under the current policy / persisted choices (a geometry not measured yet: no)."""
    if bias is not None or _env("OMNIHD_GRAD_PLANES_ONLY", "1") == "0" or torch.is_anomaly_enabled() or _fp32_policy() == "f16":
        # (anomaly mode inspects every gradient tensor: the planes-only hand-over passes bf16 planes under an fp32 view — ADVICE round 5)
        return False
    k = weight.shape[2]
    ok_f, ok_d, ok_w = conv_split_geometry(x_shape, weight.shape[0], k, stride, padding, dilation)
    if not (ok_d and ok_w and weight.requires_grad):
        return False
    pol = _fp32_policy()
    if pol == "split":
        return True
    if pol != "tune":
        return False
    geo = (tuple(x_shape), weight.shape[0], k, stride[0], padding[0], dilation[0], device_index)
    return _SPLIT_CHOICE.get(("dgrad",) + geo) == "split" and _SPLIT_CHOICE.get(("wgrad",) + geo) == "split"


def column_sums(rows2d):
    """fp32 column sums of a contiguous (rows, c) bf16 / fp32 device matrix, any c (omnihd_column_sums)."""
    if not (rows2d.is_cuda and rows2d.dim() == 2 and rows2d.is_contiguous() and rows2d.dtype in (torch.bfloat16, torch.float32)):
        raise TypeError("column_sums takes a contiguous 2-D bf16 or fp32 CUDA(HIP) tensor")
    rows, c = rows2d.shape
    dev = rows2d.device
    sums = torch.empty(c, dtype=torch.float32, device=dev)
    if rows == 0:
        return sums.zero_()
    L = lib()
    with _on(dev):
        ws = _wgrad_workspace(L.omnihd_column_sums_workspace_bytes(rows, c), dev)
        check(L.omnihd_column_sums(rows2d.data_ptr(), 1 if rows2d.dtype == torch.float32 else 0, rows, c, sums.data_ptr(),
                                   ws.data_ptr(), ws.numel(), _raw_stream()), "omnihd_column_sums")
    return sums


class _ConvBiasColsum(torch.autograd.Function):
    """y = conv2d(x, w, bias) for output widths that are not a multiple of 8 (DepthNet's 59 depth logits): the convolution and
    its data / weight gradients are torch's (MIOpen), the BIAS gradient sum_{n,h,w} g comes from the column-sum kernel — torch
    reduces such an NHWC gradient element by element (0.36 ms at 6 x 59 x 64 x 176; 10 us here).  The convolution runs INSIDE
    the function, so the result is a fresh tensor that in-place consumers (ReLU(inplace=True), sigmoid_()) may overwrite."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, groups):
        ctx.save_for_backward(x, weight)
        ctx.conv = (list(stride), list(padding), list(dilation), int(groups))
        ctx.bdtype = bias.dtype
        return torch.nn.functional.conv2d(x, weight, bias.to(x.dtype), stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.conv
        gx = gw = gb = None
        own_w = (ctx.needs_input_grad[1] and deterministic() and groups == 1 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                 and x.shape[1] % 8 == 0 and weight.shape[2] == weight.shape[3] and weight.shape[2] in (1, 3)
                 and _pair_same(stride) is not None and _pair_same(padding) is not None and _pair_same(dilation) is not None)
        if ctx.needs_input_grad[0] or (ctx.needs_input_grad[1] and not own_w):
            gx, gw, _ = torch.ops.aten.convolution_backward(g, x, weight, None, stride, padding, dilation, False, [0, 0], groups,
                                                            [bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1] and not own_w), False])
        if own_w:
            # OMNIHD_DETERMINISTIC=1: the library's fp32 weight-gradient solvers for these layers (59 depth logits, the 18 offsets
            # of the deformable convolution) accumulate with atomics — the only two gradients of the step that differed between
            # runs (profiles/round5/determinism_leftovers.txt).  Ours with the output gradient zero-padded to a multiple of 8.
            gw = wgrad_split_padded(x, g, weight.shape[2], stride[0], padding[0], dilation[0]).to(weight.dtype)
        if ctx.needs_input_grad[2]:
            gc = g if g.is_contiguous(memory_format=torch.channels_last) else g.contiguous(memory_format=torch.channels_last)
            n, c, h, w = gc.shape
            gb = column_sums(gc.permute(0, 2, 3, 1).reshape(n * h * w, c)).to(ctx.bdtype)    # a view of the NHWC memory
        return gx, gw, gb, None, None, None, None


def wgrad_split_padded(x, g, k, stride, padding, dilation):
    """fp32-grade weight gradient (Cout,Cin,k,k) of a convolution whose output channel count is NOT a multiple of 8: the output
    gradient is zero-padded to the next multiple (one small copy), the split chain runs, the padding rows are dropped."""
    cout = g.shape[1]
    cp = (cout + 7) // 8 * 8
    xc = x.float().contiguous(memory_format=torch.channels_last)
    gp = torch.empty((g.shape[0], cp, g.shape[2], g.shape[3]), dtype=torch.float32, device=g.device, memory_format=torch.channels_last)
    gp[:, cout:].zero_()
    gp[:, :cout].copy_(g)
    dw = conv_wgrad_split(split_f32(xc), split_f32(gp), int(k), int(stride), int(padding), int(dilation))
    return dw[:cout]


def conv_bias_colsum_supported(x, weight, bias):
    return (bias is not None and bias.requires_grad and torch.is_grad_enabled() and x.is_cuda and x.dim() == 4
            and x.dtype in (torch.bfloat16, torch.float32) and weight.shape[0] % 8 != 0 and weight.dtype == x.dtype)


def conv_bias_colsum(x, weight, bias, stride, padding, dilation, groups=1):
    if (deterministic() and groups == 1 and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4
            and weight.shape[2] == weight.shape[3]):
        # OMNIHD_DETERMINISTIC=1: no pass of these layers on the library (its choice of solver — and with it the last bits —
        # follows whatever tuning records the box holds; tests/test_determinism_gpu.py failed exactly when the user database was
        # seeded).  Output channels zero-padded to a multiple of 8, all three passes on the split kernels, padding cut off.
        cout = weight.shape[0]
        cp = (cout + 7) // 8 * 8
        if all(conv_split_geometry(x.shape, cp, weight.shape[2], tuple(stride), tuple(padding), tuple(dilation))):
            wp = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 0, 0, cp - cout))
            bp = torch.nn.functional.pad(bias.float(), (0, cp - cout))
            y = conv_split(x, wp, bp, tuple(stride), tuple(padding), tuple(dilation))
            return y[:, :cout].contiguous(memory_format=torch.channels_last)
    return _ConvBiasColsum.apply(x, weight, bias, tuple(stride), tuple(padding), tuple(dilation), int(groups))


class _ConvHipWgrad(torch.autograd.Function):
    """Convolution of the bf16 training path: forward and data gradient on the implicit-GEMM MFMA kernel of this library or
    on MIOpen (measured per geometry), weight gradient on the k-major MFMA chain or MIOpen (measured per geometry)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation):
        # ``weight`` / ``bias`` may be the fp32 master parameters: they are rounded to the activation dtype here
        # and their gradients are returned in THEIR dtype, so autograd adds no cast kernels of its own.
        wb = bf16_of(weight) if x.dtype == torch.bfloat16 else weight.detach().to(x.dtype)
        ctx.save_for_backward(x, wb)
        ctx.wparam = weakref.ref(weight) if (weight.dtype == torch.float32 and x.dtype == torch.bfloat16) else None
        ctx.has_bias = bias is not None
        ctx.conv = (list(stride), list(padding), list(dilation))
        ctx.param_dtypes = (weight.dtype, None if bias is None else bias.dtype)
        run_miopen = lambda: torch.nn.functional.conv2d(x, wb, None if bias is None else bias.detach().to(x.dtype), stride,
                                                         padding, dilation)
        if x.dtype == torch.bfloat16 and x.dim() == 4:
            return _conv_impl("fwd", x, wb.contiguous(memory_format=torch.channels_last), stride, padding, dilation, run_miopen,
                              None if bias is None else bias.detach(), in_shape=tuple(x.shape))
        return run_miopen()

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        stride, padding, dilation = ctx.conv
        gx = gw = gb = None
        g = g.contiguous(memory_format=torch.channels_last)
        if ctx.needs_input_grad[0]:
            run_miopen = lambda: torch.ops.aten.convolution_backward(g, x, weight, None, stride, padding, dilation, False, [0, 0],
                                                                     1, [True, False, False])[0]
            k = weight.shape[2]
            if (g.dtype == torch.bfloat16 and weight.shape[1] % 8 == 0 and weight.shape[0] % 8 == 0 and k in (1, 3)
                    and _conv_policy() != "miopen"):
                wt = lambda: bf16_dgrad_image(None if ctx.wparam is None else ctx.wparam(), weight)
                gx = _conv_impl("dgrad", g, wt, stride, padding, dilation, run_miopen, n_out=weight.shape[1], k=k,
                                in_shape=tuple(x.shape))
            else:
                gx = run_miopen()
        else:
            # no data gradient asked for (first trainable layer behind a frozen trunk): nothing to measure in this direction
            _CONV_CHOICE.setdefault(("dgrad", tuple(g.shape), weight.shape[1], weight.shape[2], dilation[0], g.device.index), "miopen")
            _CONV_CHOICE.setdefault(("dgrad_gen", tuple(x.shape), weight.shape[0], weight.shape[2], stride[0], padding[0], dilation[0],
                                     g.device.index), "miopen")
        if ctx.needs_input_grad[1]:
            gw = _tuned_wgrad(x.contiguous(memory_format=torch.channels_last), g, weight, stride, padding,
                              dilation).to(ctx.param_dtypes[0])
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(dim=(0, 2, 3), dtype=torch.float32).to(ctx.param_dtypes[1])
        return gx, gw, gb, None, None, None


def conv3x3(x, weight, bias=None):
    return _ConvHipWgrad.apply(x, weight, bias, (1, 1), (1, 1), (1, 1))


def conv_hip_wgrad(x, weight, bias, stride, padding, dilation=(1, 1)):
    return _ConvHipWgrad.apply(x, weight, bias, tuple(stride), tuple(padding), tuple(dilation))


def _gen_or_library(key, run_gen, run_lib, dev, policy):
    """The general kernel or the library's, per geometry: 'hip' / 'miopen' policies decide, 'tune' measures once (_CONV_CHOICE)."""
    if policy == "miopen":
        return run_lib()
    if policy in ("hip", "split"):
        return run_gen()
    choice = _CONV_CHOICE.get(key)
    if choice is None:
        clock = lambda fn: _clock(fn, dev, n=5, warm=2)
        choice = _CONV_CHOICE.measured(key, "hip" if clock(run_gen) <= clock(run_lib) else "miopen")
    return run_gen() if choice == "hip" else run_lib()


def _deconv_as_conv(x_shape, weight_shape, k):
    """A transposed convolution with kernel == stride k, weight (Cin_t, Cout_t, k, k), on x (B, Cin_t, H, W) IS the data gradient
    of the stride-k convolution whose weight is that tensor read as (cout = Cin_t, cin = Cout_t): returns that convolution's
    (input shape, cout)."""
    B, cin_t, H, W = x_shape
    return (B, weight_shape[1], H * k, W * k), cin_t


class _DeconvHipWgrad(torch.autograd.Function):
    """ConvTranspose2d with kernel == stride (non-overlapping up-sampling, SECONDFPN's ``deblocks``): forward and data gradient on
    the general implicit-GEMM kernel (csrc/conv_gen.hip: k*k one-tap classes in one launch / a stride-k forward) or MIOpen,
    measured per geometry; the weight gradient dW[cin][cout][ky][kx] = sum_m X[m][cin] * G[(s*y+ky, s*x+kx)][cout] is a 1x1
    weight gradient once G is viewed as rows of (ky, kx, cout) per INPUT pixel."""

    @staticmethod
    def forward(ctx, x, weight, k):
        wb = bf16_of(weight) if x.dtype == torch.bfloat16 else weight.detach().to(x.dtype)
        ctx.save_for_backward(x, wb)
        ctx.k, ctx.wdtype = k, weight.dtype
        ctx.wparam = weakref.ref(weight) if (weight.dtype == torch.float32 and x.dtype == torch.bfloat16) else None
        run_lib = lambda: torch.nn.functional.conv_transpose2d(x, wb, None, stride=k)
        conv_in, conv_cout = _deconv_as_conv(x.shape, weight.shape, k)
        if (x.dtype == torch.bfloat16 and weight.shape[1] % 8 == 0 and conv_gen_supported(1, conv_in, conv_cout, k, k, 0, 1)):
            wt = lambda: bf16_dgrad_image(None if ctx.wparam is None else ctx.wparam(), wb)
            run_gen = lambda: conv_gen(1, x.contiguous(memory_format=torch.channels_last), wt(), None, conv_in, conv_cout, k, k, 0, 1)
            return _gen_or_library(("deconv_fwd", tuple(x.shape), weight.shape[1], k, x.device.index), run_gen, run_lib, x.device,
                                   _conv_policy())
        return run_lib()

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        k = ctx.k
        gx = gw = None
        g = g.contiguous(memory_format=torch.channels_last)
        if ctx.needs_input_grad[0]:
            run_lib = lambda: torch.nn.functional.conv2d(g, weight, None, stride=k)          # adjoint of the transposed conv
            conv_in, conv_cout = _deconv_as_conv(x.shape, weight.shape, k)
            if g.dtype == torch.bfloat16 and conv_cout % 8 == 0 and conv_gen_supported(0, conv_in, conv_cout, k, k, 0, 1):
                w_cl = weight if weight.is_contiguous(memory_format=torch.channels_last) else weight.contiguous(memory_format=torch.channels_last)
                run_gen = lambda: conv_gen(0, g, w_cl, None, conv_in, conv_cout, k, k, 0, 1)
                gx = _gen_or_library(("deconv_dgrad", tuple(x.shape), weight.shape[1], k, g.device.index), run_gen, run_lib, g.device,
                                     _conv_policy())
            else:
                gx = run_lib()
        if ctx.needs_input_grad[1]:
            B, cout, Ho, Wo = g.shape
            H, W = Ho // k, Wo // k
            # (B, Ho, Wo, cout) memory -> (B, H, W, ky, kx, cout): one row of k*k*cout values per input pixel
            rows = g.permute(0, 2, 3, 1).reshape(B, H, k, W, k, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, k * k * cout)
            rows = rows.permute(0, 3, 1, 2)                                     # NCHW-shaped view of NHWC memory
            dw = conv_wgrad(x.contiguous(memory_format=torch.channels_last), rows.contiguous(memory_format=torch.channels_last),
                            1, 1, 0, 1)                                         # (k*k*cout, cin, 1, 1)
            cin = x.shape[1]
            gw = dw.reshape(k, k, cout, cin).permute(3, 2, 0, 1).to(ctx.wdtype)
        return gx, gw, None


class _DeconvSplit(torch.autograd.Function):
    """The same transposed convolution in the fp32 step: forward and data gradient on the general kernel in its fp32-grade split
    form, weight gradient on the split chain through the same space-to-depth view.  No library kernel, no atomics."""

    @staticmethod
    def forward(ctx, x, weight, k):
        x = x.contiguous(memory_format=torch.channels_last)
        xs = take_planes(x)
        if xs is None:
            xs = split_f32(x)
        ctx.save_for_backward(xs[0], xs[1], weight)
        ctx.k = k
        conv_in, conv_cout = _deconv_as_conv(x.shape, weight.shape, k)
        return conv_gen(1, xs, split_weight(weight, dgrad=True), None, conv_in, conv_cout, k, k, 0, 1)

    @staticmethod
    def backward(ctx, g):
        x_hi, x_lo, weight = ctx.saved_tensors
        k = ctx.k
        gx = gw = None
        g = g.float().contiguous(memory_format=torch.channels_last)
        conv_in, conv_cout = _deconv_as_conv(x_hi.shape, weight.shape, k)
        if ctx.needs_input_grad[0]:
            gx = conv_gen(0, split_f32(g), split_weight(weight), None, conv_in, conv_cout, k, k, 0, 1)
        if ctx.needs_input_grad[1]:
            B, cout, Ho, Wo = g.shape
            H, W = Ho // k, Wo // k
            rows = g.permute(0, 2, 3, 1).reshape(B, H, k, W, k, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, k * k * cout)
            rows = rows.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
            dw = conv_wgrad_split((x_hi, x_lo), split_f32(rows), 1, 1, 0, 1)           # (k*k*cout, cin, 1, 1)
            cin = x_hi.shape[1]
            gw = dw.reshape(k, k, cout, cin).permute(3, 2, 0, 1).to(weight.dtype)
        return gx, gw, None


def deconv_split_supported(x, weight, kernel_size, stride, padding, output_padding, groups, dilation, bias):
    k, s = _pair_same(kernel_size), _pair_same(stride)
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight.dtype == torch.float32 and k is not None and k == s
            and 1 <= k <= 4 and _pair_same(padding) == 0 and _pair_same(output_padding) == 0 and groups == 1
            and _pair_same(dilation) == 1 and bias is None and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0):
        return False
    conv_in, conv_cout = _deconv_as_conv(x.shape, weight.shape, k)
    return (conv_gen_supported(1, conv_in, conv_cout, k, k, 0, 1) and conv_gen_supported(0, conv_in, conv_cout, k, k, 0, 1)
            and _fp32_policy() != "miopen")


def deconv_split(x, weight, k):
    return _DeconvSplit.apply(x, weight, int(k))


def deconv_supported(x, weight, kernel_size, stride, padding, output_padding, groups, dilation, bias):
    k, s = _pair_same(kernel_size), _pair_same(stride)
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and k is not None and k == s and _pair_same(padding) == 0
            and _pair_same(output_padding) == 0 and groups == 1 and _pair_same(dilation) == 1 and bias is None
            and weight.shape[0] % 8 == 0 and (k * k * weight.shape[1]) % 8 == 0)


def deconv_hip_wgrad(x, weight, k):
    return _DeconvHipWgrad.apply(x, weight, int(k))


# ---------------------------------------------------------------------------------------------
# deformable 3x3 sampling (DepthNet's DCN)
# ---------------------------------------------------------------------------------------------
class _DcnSample(torch.autograd.Function):
    """x (B,H,W,C) bf16 or fp32, offset (B,Ho,Wo,18) fp32 -> col (B*Ho*Wo, 9*C) in x's type (row gathers, no atomics)."""

    @staticmethod
    def forward(ctx, x, offset, stride, pad, dil):
        B, H, W, C = x.shape
        Ho, Wo = offset.shape[1:3]
        col = torch.empty((B * Ho * Wo, 9 * C), dtype=x.dtype, device=x.device)
        sfx = "_f32" if x.dtype == torch.float32 else ""
        with _on(x.device):
            check(getattr(lib(), "omnihd_dcn3x3_sample_fwd" + sfx)(_ptr(x), _ptr(offset), _ptr(col), B, H, W, C, stride, pad,
                                                                   dil, _stream()), "omnihd_dcn3x3_sample_fwd" + sfx)
        ctx.save_for_backward(x, offset)
        ctx.geo = (stride, pad, dil)
        return col

    @staticmethod
    def backward(ctx, gcol):
        x, offset = ctx.saved_tensors
        stride, pad, dil = ctx.geo
        B, H, W, C = x.shape
        gcol = gcol.contiguous().to(x.dtype)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        goff = torch.empty_like(offset) if ctx.needs_input_grad[1] else None
        radius = offset.abs().amax().ceil().to(torch.int32).reshape(1)          # stays on the device
        sfx = "_f32" if x.dtype == torch.float32 else ""
        with _on(x.device):
            check(getattr(lib(), "omnihd_dcn3x3_sample_bwd" + sfx)(_ptr(x), _ptr(offset), _ptr(gcol), _ptr(radius), _ptr(gx),
                                                                   _ptr(goff), B, H, W, C, stride, pad, dil, _stream()),
                  "omnihd_dcn3x3_sample_bwd" + sfx)
        return gx, goff, None, None, None


def dcn3x3_sample(x_nhwc, offset_nhwc, stride=1, pad=1, dil=1):
    if not (x_nhwc.is_cuda and x_nhwc.dtype in (torch.bfloat16, torch.float32) and x_nhwc.is_contiguous()):
        raise TypeError("x must be a contiguous (B,H,W,C) bf16 or fp32 CUDA(HIP) tensor")
    if not (offset_nhwc.dtype == torch.float32 and offset_nhwc.is_contiguous() and offset_nhwc.shape[-1] == 18):
        raise TypeError("offset must be a contiguous (B,Ho,Wo,18) fp32 tensor")
    return _DcnSample.apply(x_nhwc, offset_nhwc, int(stride), int(pad), int(dil))


def dcn3x3_supported(x, k, stride, deform_groups):
    return (x.is_cuda and x.dim() == 4 and k == 3 and stride == 1 and deform_groups == 1 and x.shape[1] in (32, 64, 128, 256))


# --------------------------------------------------------------------------------------------
# Test-time post-process: rotated BEV NMS (mmdet3d v0.17.1 `nms_gpu` / `boxes_iou_bev`)
# --------------------------------------------------------------------------------------------
def nms_rotated(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """mmdet3d `nms_gpu(boxes, scores, thresh, pre_maxsize, post_max_size)`: boxes (N,5) fp32
    (x1, y1, x2, y2, ry), scores (N,) -> int64 indices of the kept boxes in descending score
    order.  Sorting stays on torch (as upstream); masks and their reduction run in the HIP library."""
    _want(boxes, torch.float32, "boxes")
    _same_device(boxes, scores)
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    sorted_boxes = boxes[order].contiguous()
    n = sorted_boxes.shape[0]
    dev = boxes.device
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    num_out = torch.zeros(1, dtype=torch.int32, device=dev)
    with _on(dev):
        ws = _workspace(lib().omnihd_nms_rotated_workspace_bytes(n), dev)
        check(lib().omnihd_nms_rotated(_ptr(sorted_boxes), n, float(thresh), _ptr(keep), _ptr(num_out), _ptr(ws),
                                       ws.numel(), _stream()), "omnihd_nms_rotated")
    kept = order[keep[:int(num_out.item())]].contiguous()
    if post_max_size is not None:
        kept = kept[:post_max_size]
    return kept


def iou_bev_matrix(boxes_a, boxes_b):
    """(Na,5) x (Nb,5) (x1,y1,x2,y2,ry) -> (Na,Nb) rotated BEV IoU (mmdet3d `boxes_iou_bev`)."""
    _want(boxes_a, torch.float32, "boxes_a")
    _want(boxes_b, torch.float32, "boxes_b")
    _same_device(boxes_a, boxes_b)
    out = torch.empty((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    with _on(boxes_a.device):
        check(lib().omnihd_iou_bev_matrix(_ptr(boxes_a), boxes_a.shape[0], _ptr(boxes_b), boxes_b.shape[0],
                                          _ptr(out), _stream()), "omnihd_iou_bev_matrix")
    return out


# --------------------------------------------------------------------------------------------
# Frozen-BatchNorm epilogue: y = act(x * scale + shift (+ residual)), channels-last bf16
# --------------------------------------------------------------------------------------------
class _AffineAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, res, relu):
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        # TF32-grade neighbours (OMNIHD_FP32_CONV=f16): y also as its half plane once a convolution has asked for it (see take_half)
        f16 = x.dtype == torch.float32 and _fp32_policy() == "f16"
        pkey = ("aff_y", scale.data_ptr()) if f16 else None
        y16 = torch.empty_like(x, dtype=torch.float16) if (pkey is not None and half_wanted(pkey)) else None
        with _on(x.device):
            if y16 is not None:
                check(lib().omnihd_affine_act_fwd_f32_planes(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), None if res is None else res.data_ptr(),
                                                             y.data_ptr(), y16.data_ptr(), None, n * h * w, c, 1 if relu else 0, _raw_stream()),
                      "omnihd_affine_act_fwd_f32_planes")
            else:
                fwd = lib().omnihd_affine_act_fwd_f32 if x.dtype == torch.float32 else lib().omnihd_affine_act_fwd
                check(fwd(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), None if res is None else res.data_ptr(), y.data_ptr(),
                          n * h * w, c, 1 if relu else 0, _raw_stream()), "omnihd_affine_act_fwd")
        ctx.save_for_backward(y if relu else None, scale)
        ctx.relu, ctx.has_res, ctx.dtype, ctx.f16 = relu, res is not None, x.dtype, f16
        if y16 is not None:
            tag_half(y, y16, pkey)
        elif pkey is not None:
            tag_producer(y, pkey)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, scale = ctx.saved_tensors
        gy = gy.to(ctx.dtype).contiguous(memory_format=torch.channels_last)
        n, c, h, w = gy.shape
        gx = torch.empty_like(gy)
        gres = torch.empty_like(gy) if ctx.has_res and ctx.needs_input_grad[3] else None
        with _on(gy.device):
            if ctx.f16 and gy.dtype == torch.float32 and f16_handover():
                # the convolution in front of this layer casts gx to half with a scale: max |gx| is accumulated here, on the way
                slot = _amax_slot(gy.device)
                check(lib().omnihd_affine_act_bwd_f32_amax(gy.data_ptr(), None if y is None else y.data_ptr(), scale.data_ptr(), gx.data_ptr(),
                                                           None if gres is None else gres.data_ptr(), slot.data_ptr(), n * h * w, c,
                                                           1 if ctx.relu else 0, _raw_stream()), "omnihd_affine_act_bwd_f32_amax")
                gx._omnihd_amax = (slot, gx._version)
            else:
                bwd = lib().omnihd_affine_act_bwd_f32 if gy.dtype == torch.float32 else lib().omnihd_affine_act_bwd
                check(bwd(gy.data_ptr(), None if y is None else y.data_ptr(), scale.data_ptr(), gx.data_ptr(),
                          None if gres is None else gres.data_ptr(), n * h * w, c, 1 if ctx.relu else 0, _raw_stream()),
                      "omnihd_affine_act_bwd")
        return gx, None, None, gres, None


def affine_act_supported(x, res=None):
    ok = x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float32) and x.shape[1] % 8 == 0
    return ok and (res is None or (res.shape == x.shape and res.dtype == x.dtype and res.is_cuda))


def affine_act(x, scale, shift, res=None, relu=True):
    """x, res (N,C,H,W) bf16 or fp32 (made channels-last if they are not), scale/shift (C,) fp32 constants."""
    x = x.contiguous(memory_format=torch.channels_last)
    if res is not None:
        res = res.contiguous(memory_format=torch.channels_last)
    _want(scale, torch.float32, "scale")
    _want(shift, torch.float32, "shift")
    return _AffineAct.apply(x, scale, shift, res, relu)


# --------------------------------------------------------------------------------------------
# Training-mode BatchNorm (+ReLU), statistics optionally averaged over ranks (naive SyncBN)
# --------------------------------------------------------------------------------------------
def _rows_view(t):
    """(N,C,H,W) channels-last or (N,C) contiguous bf16 -> (rows, c)."""
    if t.dim() == 4:
        return t.shape[0] * t.shape[2] * t.shape[3], t.shape[1]
    return t.shape[0], t.shape[1]


def _f32c(t):
    # only the data pointer is read: a contiguous fp32 parameter is used as it is (no detach() object per call)
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().float().contiguous()


class _BnTrainAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, group, res, unbiased_sync=False,
                grad_planes_only=False):
        import torch.distributed as dist
        ctx.grad_planes_only = bool(grad_planes_only)
        rows, c = _rows_view(x)
        dev = x.device
        ranks = dist.get_world_size(group) if group is not None else 1
        sfx = "_f32" if x.dtype == torch.float32 else ""
        buf = torch.empty(6, c, dtype=torch.float32, device=dev)           # [0:2] statistics, [2:6] scale, shift, mean, invstd
        stats, consts = buf[:2].view(-1), buf[2:]
        y = torch.empty_like(x)
        gamma, beta = _f32c(weight), _f32c(bias)
        L = lib()
        st = _raw_stream()
        # fp32 4-D outputs can be handed to the next split convolution as planes (see take_planes)
        pkey = ("bn_y", id(weight)) if (x.dtype == torch.float32 and x.dim() == 4 and ranks == 1) else None
        y_planes = _alloc_planes(y) if (pkey is not None and planes_wanted(pkey)) else None
        if y_planes is None and pkey is not None and half_wanted(pkey):
            y_planes = (torch.empty_like(y, dtype=torch.float16),)      # TF32-grade neighbour: the half plane of y (see take_half)
        rm = None if running_mean is None else running_mean.data_ptr()
        rv = None if running_var is None else running_var.data_ptr()
        resp = None if res is None else res.data_ptr()
        with _on(dev):
            nbytes = _SIZE_CACHE.get(("bn", rows, c))
            if nbytes is None:
                nbytes = _SIZE_CACHE[("bn", rows, c)] = L.omnihd_bn_workspace_bytes(rows, c)
            ws = _wgrad_workspace(nbytes, dev)
            if ranks == 1:
                # torch's BatchNorm keeps the unbiased variance in running_var
                corr = rows / (rows - 1.0) if rows > 1 else 1.0
                if y_planes is not None:
                    check(L.omnihd_bn_train_fwd_f32_planes(
                        x.data_ptr(), resp, gamma.data_ptr(), beta.data_ptr(), rm, rv, momentum, eps, corr, 1 if relu else 0,
                        y.data_ptr(), y_planes[0].data_ptr(), y_planes[1].data_ptr() if len(y_planes) == 2 else None, stats.data_ptr(),
                        consts.data_ptr(), rows, c, ws.data_ptr(), ws.numel(), st), "omnihd_bn_train_fwd_f32_planes")
                else:
                    check(getattr(L, "omnihd_bn_train_fwd" + sfx)(
                        x.data_ptr(), resp, gamma.data_ptr(), beta.data_ptr(), rm, rv, momentum, eps, corr, 1 if relu else 0,
                        y.data_ptr(), stats.data_ptr(), consts.data_ptr(), rows, c, ws.data_ptr(), ws.numel(), st), "omnihd_bn_train_fwd")
            else:
                check(getattr(L, "omnihd_bn_channel_sums" + sfx)(x.data_ptr(), None, None, None, stats.data_ptr(), rows, c, 0,
                                                                 1.0 / rows, ws.data_ptr(), ws.numel(), st), "omnihd_bn_channel_sums")
                dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
                # the reference's SyncBN keeps the biased variance (ops/norm.py:74-75); torch's SyncBatchNorm the unbiased
                # one over the rows of all ranks
                total = float(ranks * rows)
                corr = total / (total - 1.0) if (unbiased_sync and total > 1) else 1.0
                check(L.omnihd_bn_fwd_consts(stats.data_ptr(), 1.0 / ranks, gamma.data_ptr(), beta.data_ptr(), eps, momentum, corr,
                                             c, rm, rv, consts[0].data_ptr(), consts[1].data_ptr(), consts[2].data_ptr(),
                                             consts[3].data_ptr(), st), "omnihd_bn_fwd_consts")
                check(getattr(L, "omnihd_affine_act_fwd" + sfx)(x.data_ptr(), consts[0].data_ptr(), consts[1].data_ptr(), resp,
                                                                y.data_ptr(), rows, c, 1 if relu else 0, st), "omnihd_affine_act_fwd")
        # The ReLU mask of the backward comes from the saved output.  The kernels can also recompute it from x with the
        # forward's constants (OMNIHD_BN_MASK_FROM_X=1: one tensor less to read), but that measured SLOWER in the full
        # step (34.1-35.0 vs 32.5-33.3 ms, alternating blocks in one process): the per-element constant loads cost more
        # than the streamed read they save.
        keep_y = relu and (res is not None or _env("OMNIHD_BN_MASK_FROM_X", "0") != "1")
        ctx.save_for_backward(x, y if keep_y else None, gamma, consts)
        ctx.relu, ctx.group, ctx.ranks, ctx.param_dtypes = relu, group, ranks, (weight.dtype, bias.dtype)
        ctx.has_res = res is not None
        ctx.gkey = ("bn_gx", id(weight)) if pkey is not None else None
        if pkey is not None:
            if y_planes is not None and len(y_planes) == 1:
                tag_half(y, y_planes[0], pkey)
            elif y_planes is not None:
                tag_planes(y, y_planes, pkey)
            else:
                tag_producer(y, pkey)
        return y

    @staticmethod
    def backward(ctx, gy):
        import torch.distributed as dist
        x, y, gamma, consts = ctx.saved_tensors
        gy = gy.to(x.dtype)
        gy = gy.contiguous(memory_format=torch.channels_last) if gy.dim() == 4 else gy.contiguous()
        sfx = "_f32" if x.dtype == torch.float32 else ""
        rows, c = _rows_view(x)
        dev = x.device
        buf = torch.empty(7, c, dtype=torch.float32, device=dev)           # [0:2] sums, [2:7] dgamma, dbeta, A, B, C
        local, out = buf[:2].view(-1), buf[2:]
        gx = torch.empty_like(x)
        gres = None
        if ctx.has_res and ctx.needs_input_grad[9]:
            gres = torch.empty_like(x) if ctx.relu else gy          # without a ReLU the residual's gradient is gy itself
        gresp = gres.data_ptr() if (gres is not None and ctx.relu) else None
        yp = None if y is None else y.data_ptr()
        L = lib()
        st = _raw_stream()
        with _on(dev):
            ws = _wgrad_workspace(_SIZE_CACHE[("bn", rows, c)], dev)
            if ctx.grad_planes_only and ctx.ranks == 1 and x.dtype == torch.float32 and x.numel() % 8 == 0:
                # The convolution in front of this layer is the ONLY consumer of gx (conv_bn_act keeps the tensor between them
                # to itself) and reads it as hi / lo planes: write the planes only.  The fp32-typed tensor autograd carries
                # between the two nodes is a view of the plane buffer (same byte count) — its fp32 values are never read.
                n = x.numel()
                buf = torch.empty(2 * n, dtype=torch.bfloat16, device=dev)
                gx_planes = tuple(buf[i * n:(i + 1) * n].as_strided(x.shape, x.stride()) for i in (0, 1))
                gx = buf.view(torch.float32).as_strided(x.shape, x.stride())
                check(L.omnihd_bn_train_bwd_f32_planes(
                    gy.data_ptr(), yp, 1 if ctx.relu else 0, x.data_ptr(), gamma.data_ptr(), consts.data_ptr(), None,
                    gx_planes[0].data_ptr(), gx_planes[1].data_ptr(), gresp, local.data_ptr(), out.data_ptr(), rows, c,
                    ws.data_ptr(), ws.numel(), st), "omnihd_bn_train_bwd_f32_planes")
                gx._omnihd_planes = (gx_planes, gx._version, ("bn_gx_only", 0))
                gx._omnihd_planes_only = True
                FAST_PATHS["grad_planes_only"] = FAST_PATHS.get("grad_planes_only", 0) + 1
                return (gx, out[0].to(ctx.param_dtypes[0]), out[1].to(ctx.param_dtypes[1]), None, None, None, None, None, None,
                        gres, None, None)
            gx_planes = _alloc_planes(gx) if (ctx.gkey is not None and ctx.ranks == 1 and planes_wanted(ctx.gkey)) else None
            if gx_planes is not None:
                check(L.omnihd_bn_train_bwd_f32_planes(
                    gy.data_ptr(), yp, 1 if ctx.relu else 0, x.data_ptr(), gamma.data_ptr(), consts.data_ptr(), gx.data_ptr(),
                    gx_planes[0].data_ptr(), gx_planes[1].data_ptr(), gresp, local.data_ptr(), out.data_ptr(), rows, c,
                    ws.data_ptr(), ws.numel(), st), "omnihd_bn_train_bwd_f32_planes")
                tag_planes(gx, gx_planes, ctx.gkey)
            elif ctx.ranks == 1 and ctx.gkey is not None and f16_handover():
                # the TF32-grade convolution in front of this layer casts gx to half with a scale: max |gx| accumulated on the way
                slot = _amax_slot(dev)
                check(L.omnihd_bn_train_bwd_f32_amax(
                    gy.data_ptr(), yp, 1 if ctx.relu else 0, x.data_ptr(), gamma.data_ptr(), consts.data_ptr(), gx.data_ptr(),
                    slot.data_ptr(), gresp, local.data_ptr(), out.data_ptr(), rows, c, ws.data_ptr(), ws.numel(), st),
                    "omnihd_bn_train_bwd_f32_amax")
                gx._omnihd_amax = (slot, gx._version)
            elif ctx.ranks == 1:
                if ctx.gkey is not None:
                    tag_producer(gx, ctx.gkey)
                check(getattr(L, "omnihd_bn_train_bwd" + sfx)(
                    gy.data_ptr(), yp, 1 if ctx.relu else 0, x.data_ptr(), gamma.data_ptr(), consts.data_ptr(), gx.data_ptr(),
                    gresp, local.data_ptr(), out.data_ptr(), rows, c, ws.data_ptr(), ws.numel(), st), "omnihd_bn_train_bwd")
            else:
                fss = consts.data_ptr() if (ctx.relu and yp is None) else None
                check(getattr(L, "omnihd_bn_channel_sums" + sfx)(gy.data_ptr(), x.data_ptr(), yp, fss, local.data_ptr(), rows, c,
                                                                 1, 1.0, ws.data_ptr(), ws.numel(), st), "omnihd_bn_channel_sums")
                glob = local.clone()
                dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=ctx.group)
                check(L.omnihd_bn_bwd_consts(local.data_ptr(), glob.data_ptr(), gamma.data_ptr(), consts[2].data_ptr(),
                                             consts[3].data_ptr(), 1.0 / (ctx.ranks * rows), c, out[0].data_ptr(),
                                             out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), out[4].data_ptr(), st),
                      "omnihd_bn_bwd_consts")
                check(getattr(L, "omnihd_bn_bwd_apply" + sfx)(gy.data_ptr(), yp, fss, x.data_ptr(), out[2].data_ptr(),
                                                              out[3].data_ptr(), out[4].data_ptr(), gx.data_ptr(), gresp, rows, c,
                                                              st), "omnihd_bn_bwd_apply")
        return (gx, out[0].to(ctx.param_dtypes[0]), out[1].to(ctx.param_dtypes[1]), None, None, None, None, None, None,
                gres, None, None)


def bn_train_supported(x):
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() in (2, 4) and x.shape[1] % 8 == 0
            and x.shape[1] <= 2048):
        return False
    return x.numel() > 0


def bn_train_act(x, weight, bias, running_mean, running_var, momentum, eps, relu=False, group=None, residual=None,
                 unbiased_sync=False, grad_planes_only=False):
    """``act(BatchNorm_train(x) + residual)`` of a bf16 or fp32 (N,C,H,W) [made channels-last] or (N,C) tensor; statistics
    are the mean over ``group``'s ranks of the per-rank mean / mean of squares when a group with more than one rank is
    given (``unbiased_sync``: running_var takes the unbiased variance over all ranks' rows, as torch's SyncBatchNorm)."""
    cl = (lambda t: t.contiguous(memory_format=torch.channels_last)) if x.dim() == 4 else (lambda t: t.contiguous())
    return _BnTrainAct.apply(cl(x), weight, bias, running_mean, running_var, float(momentum), float(eps), bool(relu), group,
                             None if residual is None else cl(residual), bool(unbiased_sync), bool(grad_planes_only))


# --------------------------------------------------------------------------------------------
# Anchor target assignment + detection losses, fused (csrc/anchor_loss.hip)
# --------------------------------------------------------------------------------------------
class _AnchorLoss(torch.autograd.Function):
    """(cls_score, bbox_pred, dir_pred) -> (loss_cls, loss_bbox, loss_dir) of Anchor3DHead.loss for one feature level: target
    assignment, the three losses and the (unscaled) gradients of the three maps in three launches; the backward scales the saved
    gradient maps in one.  See include/omnihd_hip.h: omnihd_anchor_loss_fwd."""

    @staticmethod
    def forward(ctx, cls_score, bbox_pred, dir_pred, anchors, gt_boxes, gt_labels, gt_offsets, meta):
        (num_classes, code_size, na, params7, sin_diff, code_weight, loss_weights) = meta
        # (fp32 and dense in NCHW or NHWC memory: ``anchor_loss`` casts / copies OUTSIDE the function where needed)
        B, _, H, W = cls_score.shape
        dev = cls_score.device
        total_gt = int(gt_boxes.shape[0])
        g_cls, g_box, g_dir = torch.empty_like(cls_score), torch.empty_like(bbox_pred), torch.empty_like(dir_pred)
        assert g_cls.stride() == cls_score.stride() and g_box.stride() == bbox_pred.stride() and g_dir.stride() == dir_pred.stride()
        out = torch.empty(4 + B, dtype=torch.float32, device=dev)
        strides = (ctypes.c_longlong * 12)(*cls_score.stride(), *bbox_pred.stride(), *dir_pred.stride())
        h_par = (ctypes.c_float * 7)(*[float(v) for v in params7])
        h_cw = (ctypes.c_float * code_size)(*[float(v) for v in code_weight])
        h_lw = (ctypes.c_float * 3)(*[float(v) for v in loss_weights])
        L = lib()
        with _on(dev):
            ws = _workspace(L.omnihd_anchor_loss_workspace_bytes(B, H * W * na, total_gt), dev)
            check(L.omnihd_anchor_loss_fwd(_ptr(anchors), _ptr(gt_boxes) if total_gt else None, _ptr(gt_labels) if total_gt else None,
                                           _ptr(gt_offsets), total_gt, _ptr(cls_score), _ptr(bbox_pred), _ptr(dir_pred), B, H, W, na,
                                           num_classes, code_size, ctypes.cast(strides, ctypes.c_void_p), ctypes.cast(h_par, ctypes.c_void_p),
                                           1 if sin_diff else 0, ctypes.cast(h_cw, ctypes.c_void_p), ctypes.cast(h_lw, ctypes.c_void_p),
                                           _ptr(g_cls), _ptr(g_box), _ptr(g_dir), _ptr(out), _ptr(ws), ws.numel(), _raw_stream()),
                  "omnihd_anchor_loss_fwd")
        ctx.save_for_backward(g_cls, g_box, g_dir, out)
        ctx.loss_weights = tuple(float(v) for v in loss_weights)
        info = out[3:]
        ctx.mark_non_differentiable(info)
        return out[0], out[1], out[2], info

    @staticmethod
    def backward(ctx, up_cls, up_box, up_dir, _up_info):
        if getattr(ctx, "consumed", False):
            # the gradient maps are scaled IN PLACE below (raw pointers: autograd's version counters do not see it): a second
            # backward over the same graph would scale them twice and return wrong gradients without an error
            raise RuntimeError("omnihd anchor loss: a second backward pass over the same forward is not supported (the saved "
                               "gradient maps are consumed in place); run the forward again, or set OMNIHD_ANCHOR_LOSS=0")
        ctx.consumed = True
        g_cls, g_box, g_dir, out = ctx.saved_tensors
        dev = g_cls.device
        h_lw = (ctypes.c_float * 3)(*ctx.loss_weights)
        ups = [None if u is None else u.to(torch.float32).contiguous() for u in (up_cls, up_box, up_dir)]
        with _on(dev):
            check(lib().omnihd_anchor_loss_bwd(_ptr(g_cls), g_cls.numel(), _ptr(g_box), g_box.numel(), _ptr(g_dir), g_dir.numel(),
                                               None if ups[0] is None else _ptr(ups[0]), None if ups[1] is None else _ptr(ups[1]),
                                               None if ups[2] is None else _ptr(ups[2]), _ptr(out), ctypes.cast(h_lw, ctypes.c_void_p),
                                               _raw_stream()), "omnihd_anchor_loss_bwd")
        return g_cls, g_box, g_dir, None, None, None, None, None


def anchor_loss(cls_score, bbox_pred, dir_pred, anchors, gt_boxes, gt_labels, gt_offsets, num_classes, code_size, anchors_per_loc,
                pos_iou_thr, neg_iou_thr, min_pos_iou, gamma, alpha, beta, dir_offset, sin_diff, code_weight, loss_weights):
    """Fused Anchor3DHead loss (one feature level).  gt_boxes (total, code_size) fp32 / gt_labels (total,) int32 concatenated over
    the batch, gt_offsets (B+1,) int32 on the device.  Returns (loss_cls, loss_bbox, loss_dir, info) with info = [avg_factor,
    positives per sample...]."""
    meta = (int(num_classes), int(code_size), int(anchors_per_loc),
            (pos_iou_thr, neg_iou_thr, min_pos_iou, gamma, alpha, beta, dir_offset), bool(sin_diff), tuple(code_weight), tuple(loss_weights))
    dense = lambda t: t if (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)) else t.contiguous()
    maps = [dense(t.float()) for t in (cls_score, bbox_pred, dir_pred)]          # differentiable casts: bf16 maps get bf16 gradients
    return _AnchorLoss.apply(maps[0], maps[1], maps[2], anchors, gt_boxes, gt_labels, gt_offsets, meta)


# --------------------------------------------------------------------------------------------
# Radar sweep merge on the device (LoadRadarPointsMultiSweeps arithmetic)
# --------------------------------------------------------------------------------------------
def radar_merge(raw, sweep_offsets, sweep_consts, pc_range=None):
    """raw (N, load_dim) fp32, sweep_offsets (S+1,) int32, sweep_consts (S, 17) fp64 -> (points (N, 10) fp32,
    in_range (N,) bool or None).  See include/omnihd_hip.h: omnihd_radar_merge."""
    _want(raw, torch.float32, "raw"); _want(sweep_offsets, torch.int32, "sweep_offsets")
    _want(sweep_consts, torch.float64, "sweep_consts")
    dev = _same_device(raw, sweep_offsets, sweep_consts)
    n, load_dim = raw.shape
    n_sweeps = sweep_offsets.numel() - 1
    if sweep_consts.shape != (n_sweeps, 17):
        raise ValueError("sweep_consts must be (n_sweeps, 17)")
    out = torch.empty((n, 10), dtype=torch.float32, device=dev)
    mask = rng = None
    if pc_range is not None:
        mask = torch.empty((n,), dtype=torch.uint8, device=dev)
        rng = torch.tensor([float(v) for v in pc_range], dtype=torch.float32, device=dev)
    with _on(dev):
        check(lib().omnihd_radar_merge(_ptr(raw), n, load_dim, _ptr(sweep_offsets), n_sweeps, _ptr(sweep_consts), _ptr(rng),
                                       _ptr(out), _ptr(mask), _stream()), "omnihd_radar_merge")
    return out, (None if mask is None else mask.bool())
