"""Radar branch: hard voxelisation, pillar scatter, fused pillar feature net, sweep merge (csrc/voxelize.hip, pillar_scatter.hip, pillar_pfn.hip, radar_merge.hip).
(Part of omnihd_amd.ops — the tensor-level wrappers over the C ABI; `from omnihd_amd import ops` exposes every name.)"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from .._env import env as _env
from .._lib import check, lib
from ._core import _on, _ptr, _raw_stream, _same_device, _stream, _want, _workspace



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
