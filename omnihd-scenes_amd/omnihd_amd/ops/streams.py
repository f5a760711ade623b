"""Weight gradients on a side stream (plain and under DistributedDataParallel bucket views) and the live fast-path report.
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
from ._core import FAST_PATHS
from .policy import _CHOICE_INFO
from .planes import HANDOVER_STATS
from .weights import WIMG_STATS



def fast_paths_report():
    """What ran, from live counters — not from the environment switches: the kept pooling buffers and the weight-gradient side
    stream rest on private torch hooks (``torch._C._storage_Use_Count``, ``torch._C._current_graph_task_id`` +
    ``queue_callback``) that are probed and fall back silently when a torch build lacks them."""
    from .. import plan as _plan
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
            "weight_table_uploads": int(WIMG_STATS["miss"]),
            # operand planes written by the producer's epilogue instead of a split / cast pass of the consuming convolution
            "planes_handed_over": {"split": int(HANDOVER_STATS["taken"]), "half": int(HANDOVER_STATS.get("taken_half", 0))}}


def fast_paths_reset():
    from .. import plan as _plan
    for d in (FAST_PATHS, _plan.FAST_PATHS, WIMG_STATS, HANDOVER_STATS):
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
