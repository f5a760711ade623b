"""Weight images (bf16, split, half; forward and data-gradient layouts) kept current behind the optimiser step with one launch.
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
from ._core import _on, _raw_stream
from .conv_kernels import conv_dgrad_weights
from .planes import cast_f16, split_f32



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


# ---------------------------------------------------------------------------------------------
# TF32-grade form of the fp32 step's convolutions (round 6; OMNIHD_FP32_CONV=f16): ONE half MFMA product per fp32 product
# ---------------------------------------------------------------------------------------------
# The reference trains with TF32 left on (tools/train.py:150-153): 11 significant bits per operand.  An IEEE half has the same 11
# bits; activations and weights are converted as they are, gradients with an exact power-of-two scale found per tensor (amax pass)
# whose inverse the consuming kernel applies.  Layers the half kernels do not take (strided, transposed, narrow) stay on the
# fp32-grade split kernels, so every layer of the step is at least TF32-grade.  Parity: tests/test_conv_f16_gpu.py.
_F16_SHADOW = {}


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
