"""Which implementation runs a convolution: the persisted / measured choice tables and the environment policies (OMNIHD_CONV_POLICY, OMNIHD_FP32_CONV).
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
from ._core import _current_device, deterministic
from .conv_kernels import conv_fwd, conv_fwd_supported, conv_gen, conv_gen_supported, conv_split_geometry, conv_wgrad



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


# omnihd-scenes_amd/ (this file: omnihd-scenes_amd/omnihd_amd/ops/policy.py): kernel_choices/ lies beside the package
_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _persisted_choices():
    if not _CHOICE_INFO["loaded"]:
        _CHOICE_INFO["loaded"] = True
        path = _env("OMNIHD_CHOICE_TABLE")
        if path is None:
            path = os.path.join(_PKG_ROOT, "kernel_choices", "gfx950.json")
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
        root = os.path.dirname(_PKG_ROOT)
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


def f16_handover():
    """TF32-grade policy with the producers' hand-over on (OMNIHD_F16_HANDOVER=0: every convolution runs its own cast / amax passes —
    the A/B switch of tests/test_conv_f16_gpu.py and of DESIGN.md 4.6b's numbers)."""
    return _fp32_policy() == "f16" and _env("OMNIHD_F16_HANDOVER", "1") != "0"


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
