"""Convolution kernels behind the C ABI: weight gradients, implicit-GEMM forward / data gradient (bf16, split, half), the general strided kernel, column sums.
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
from ._core import _SIZE_CACHE, _f32c, _on, _pair_same, _ptr, _raw_stream, _want_cl, _wgrad_workspace



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
