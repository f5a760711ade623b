"""The bf16 step's convolutions and transposed convolutions as autograd Functions, bias gradients by column sums.
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
from ._core import _pair_same, deterministic
from .conv_kernels import column_sums, conv_gen, conv_gen_supported, conv_split_geometry, conv_wgrad, conv_wgrad_split
from .policy import _CONV_CHOICE, _clock, _conv_impl, _conv_policy, _fp32_policy, _tuned_wgrad
from .planes import split_f32, take_planes
from .weights import bf16_dgrad_image, bf16_of, split_weight
from .conv_fp32 import conv_split



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
