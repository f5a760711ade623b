"""Operand planes of fp32 activations: the hi / lo bf16 split, the IEEE-half cast with its device-side scale, and the producer -> consumer hand-over tags.
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
from ._core import _on, _ptr, _raw_stream
from .policy import f16_handover



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
# scripts/lab/step_times.py) in round 3: 50.9 ms with, 50.6 ms without — the BatchNorm kernels' extra 4 B/element of stores cost
# what the convolutions' split passes saved, and it stayed OFF.  Round 6, the same A/B on the step as it is now (one stream; where
# the step is bound by the in-order queue a launch less is worth more than its bytes): 45.09 / 45.11 ms without, 44.83 / 44.62 ms
# with (alternating fresh processes on one box, gpurun r6_29) — ON by default (OMNIHD_SPLIT_HANDOVER=0 turns it off; results are
# bit-identical: tests/test_conv_split_gpu.py::test_batchnorm_hands_its_planes_to_the_next_split_convolution).
_PLANES_WANTED = set()
_PLANES_UNUSED = {}
HANDOVER_STATS = {"taken": 0, "stale": 0, "asked": 0, "untagged": 0}


def planes_wanted(key):
    return key in _PLANES_WANTED and _env("OMNIHD_SPLIT_HANDOVER", "1") != "0"


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
        if _env("OMNIHD_SPLIT_HANDOVER", "1") != "0":
            _PLANES_WANTED.add(key)
        HANDOVER_STATS["asked"] += 1
        return None
    if version != t._version or planes[0].shape != t.shape or planes[0].stride() != t.stride():
        HANDOVER_STATS["stale"] += 1
        return None
    _PLANES_UNUSED[key] = 0
    HANDOVER_STATS["taken"] += 1
    return planes


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


def _f16_plane(t, mode, scratch, L, st):
    """cast_f16 without its argument checks (the caller holds a dense fp32 device tensor): one allocation, one library call."""
    out = torch.empty_like(t, dtype=torch.float16)              # preserve_format: a dense tensor keeps its strides
    check(L.omnihd_cast_f16(t.data_ptr(), t.numel(), mode, out.data_ptr(), None if scratch is None else scratch.data_ptr(), st),
          "omnihd_cast_f16")
    return out
