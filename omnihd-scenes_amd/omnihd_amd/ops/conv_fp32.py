"""The fp32 step's convolutions as autograd Functions: fp32-grade split form (_ConvSplit) and TF32-grade half form (_ConvF16).
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
from ._core import FAST_PATHS, _CL, _SIZE_CACHE, _f32c, _on, _raw_stream, _wgrad_workspace
from .conv_kernels import _conv_timed, column_sums, conv_fwd_split, conv_gen, conv_split_geometry, conv_wgrad_split
from .policy import _SPLIT_CHOICE, _fp32_policy, _split_pick
from .planes import _amax_slot, _f16_plane, split_f32, take_half, take_planes
from .weights import f16_weight, split_weight
from .streams import (_DDP, _VIEW_WRITTEN, _WGRAD_ENGINE_OK, _ddp_bucket_view, _view_writable, _wgrad_pass_begin, _wgrad_side_stream)



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


def conv_f16_applies(x_shape, weight, stride, padding, dilation):
    """The half kernels take the layer in all three directions: stride-1 'same' 1x1 / 3x3, Cin and Cout multiples of 64."""
    if weight.dim() != 4 or weight.shape[2] != weight.shape[3] or weight.dtype != torch.float32:
        return False
    ok_f, ok_d, ok_w = conv_split_geometry(x_shape, weight.shape[0], weight.shape[2], stride, padding, dilation)
    return ok_f == "igemm" and ok_d == "igemm" and bool(ok_w)


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
