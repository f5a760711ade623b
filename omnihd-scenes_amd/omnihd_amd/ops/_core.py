"""Shared plumbing of the operator wrappers: raw stream handle, device guard, argument checks, workspaces, live counters.
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib



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


def _pair_same(v):
    v = tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    return v[0] if len(v) == 2 and v[0] == v[1] else None


def deterministic():
    """OMNIHD_DETERMINISTIC=1: every convolution pass this library has a kernel for runs on it (no per-geometry race against the
    library kernels, whose fp32 solvers for strided layers and small weight gradients accumulate with atomics), so that a
    training step is run-to-run identical bit for bit (tests/test_determinism_gpu.py)."""
    return _env("OMNIHD_DETERMINISTIC", "0") == "1"


# counters of the optional fast paths actually taken in this process (bench.py: `fast_paths`)
FAST_PATHS = {"wgrad_side_stream": 0, "wgrad_in_line": 0, "dual_stream_forward": 0, "single_stream_forward": 0}


_CL = torch.channels_last


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
