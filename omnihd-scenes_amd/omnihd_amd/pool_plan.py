"""Pooling plans built ON THE DEVICE, one per camera calibration, without a host round trip (csrc/pool_plan.hip).

The reference rebuilds its rank tables in every forward (bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py:283-300 ->
``voxel_pooling_prepare_v2`` :302-362): its ``lidar2img`` is composed per sample from the ego poses
(datasets/newscenes_dataset.py:203-216), so on its own frames every step sees a calibration it has not seen before.
``omnihd_amd.plan.build_plan`` answers a new calibration with two sorts, host-side scheduling loops and several device
synchronisations; ``build_device_plan`` below answers it with ONE library call that enqueues ~25 launches on the current stream and
returns at once.  What the host never learns — how many frustum points survive, how many output rows are non-empty, how many tiles
the forward is cut into — stays in ``plan.hdr`` on the device; buffers and grids are sized by upper bounds.  The counts travel to
pinned host memory asynchronously: a plan that is used again later (static rig, plan cache) launches exact grids.

The tables are the ones ``omnihd_amd.plan`` documents (direct forward, packed patch backward); results are bit-identical to a
host-built plan of the same calibration (tests/test_device_plan_gpu.py).
"""
import ctypes
import weakref

import numpy as np
import torch

from . import ops, plan as _plan
from ._env import env as _env
from ._lib import check, lib

_ptr = ops._ptr


def patch_walk(n_img, fH, fW):
    """Static walk of the patch backward over the 16-pixel patches of a frustum shape: (walk, band), both int32 [n_patch].
    Patch p = pixels [16*(p % ppi), +16) of image p // ppi (flat pixel index inside the image); the walk goes image by image in
    bands of 4 image rows, left to right inside a band — the order of ``omnihd_amd.plan.patch_schedule``."""
    fhw = fH * fW
    ppi = (fhw + _plan.PATCH - 1) // _plan.PATCH
    p = torch.arange(n_img * ppi)
    img, k = p // ppi, p % ppi
    h, w = (k * _plan.PATCH) // fW, (k * _plan.PATCH) % fW
    nb = (fH + 3) // 4
    key = ((img * nb + h // 4) * ((fW + _plan.PATCH - 1) // _plan.PATCH + 1) + w // _plan.PATCH) * 4 + h % 4
    order = torch.argsort(key, stable=True)
    band = (img * nb + h // 4)[order]
    return p[order].int().contiguous(), band.int().contiguous()


_WALKS = {}


def _walk_on(device, n_img, fH, fW):
    key = (str(device), n_img, fH, fW)
    got = _WALKS.get(key)
    if got is None:
        walk, band = patch_walk(n_img, fH, fW)
        got = _WALKS[key] = (walk.to(device), band.to(device))
    return got


_SIZES = {}


def plan_sizes(n_total, n_rows, n_pix, fhw):
    """(workspace bytes, tile capacity, patches, patch slots per XCD) of a plan (host-side arithmetic + rocPRIM size queries)."""
    key = (n_total, n_rows, n_pix, fhw)
    got = _SIZES.get(key)
    if got is None:
        out = (ctypes.c_longlong * 4)()
        check(lib().omnihd_pool_plan_sizes(n_total, n_rows, n_pix, fhw, _plan.TILE_ITEMS, _plan.LONG_LEN,
                                           ctypes.cast(out, ctypes.c_void_p)), "omnihd_pool_plan_sizes")
        got = _SIZES[key] = tuple(int(v) for v in out)
    return got


HDR_POINTS, HDR_ROWS, HDR_TILES, HDR_TILES_PER_XCD, HDR_PATCH_RUN, HDR_STATUS = range(6)


class DevicePoolPlan:
    """Tables of one calibration, resident on the device (see the module docstring).  Immutable once built."""

    def __init__(self, layout, grid, frustum, device):
        self.layout, self.grid = layout, tuple(grid)               # 'byxz' | 'bzyx', (B, Z, Y, X)
        B, Z, Y, X = self.grid
        self.n_rows = B * Z * Y * X
        self.n_img, self.depth_bins, self.fH, self.fW = frustum     # B*N, D, fH, fW
        self.feat_hw = self.fH * self.fW
        self.n_total = self.n_img * self.depth_bins * self.feat_hw
        self.n_pix = self.n_img * self.feat_hw
        self.device = device
        ws, self.tiles_cap, self.n_patch, self.patch_per = plan_sizes(self.n_total, self.n_rows, self.n_pix, self.feat_hw)
        self.workspace_bytes = ws
        i32 = dict(dtype=torch.int32, device=device)
        self.pt = torch.empty(self.n_total, **i32)
        self.ivl_rel = torch.empty(self.n_rows, **i32)
        self.desc32 = torch.empty((self.tiles_cap, 32), **i32)
        self.row_ptr = torch.empty(self.n_rows + 1, **i32)
        self.row_bin = torch.empty(self.n_total, **i32)
        self.pix_ptr = torch.empty(self.n_pix + 1, **i32)
        self.patch_order = torch.empty(8 * self.patch_per, **i32)
        self.hdr = torch.zeros(32, **i32)
        self.rows_sorted = None          # kept only on request (tests, reference-format export)
        self.ranks_depth_sorted = None
        self._hdr_host = None
        self._hdr_event = None
        self._counts = None

    # ---- the counts, once they have reached the host -----------------------------------------------------------------
    def counts(self, wait=False):
        """{points, rows, tiles, tiles_per_xcd, patch_run, status} as host integers, or None while the asynchronous copy
        that followed the build is still in flight (``wait=True`` blocks on ITS event only)."""
        if self._counts is None and self._hdr_event is not None:
            if wait:
                self._hdr_event.synchronize()
            if wait or self._hdr_event.query():
                h = self._hdr_host.tolist()
                self._counts = dict(points=h[HDR_POINTS], rows=h[HDR_ROWS], tiles=h[HDR_TILES],
                                    tiles_per_xcd=h[HDR_TILES_PER_XCD], patch_run=h[HDR_PATCH_RUN], status=h[HDR_STATUS])
                if self._counts["status"] != 0:
                    raise RuntimeError(f"device pooling plan reported status {self._counts['status']} "
                                       "(tile capacity exceeded: a bug in omnihd_pool_plan_sizes)")
                self._hdr_host = self._hdr_event = None
                _LAST_COUNTS[self._shape_key()] = self._counts
        return self._counts

    @property
    def n_points(self):
        """Number of frustum points inside the grid (blocks until the build has finished: tests and reports only)."""
        return self.counts(wait=True)["points"]

    def launch_slots(self):
        c = self.counts()
        return 8 * max(1, c["tiles_per_xcd"]) if c is not None else self.tiles_cap

    def _shape_key(self):
        return (str(self.device), self.layout, self.grid, self.n_img, self.depth_bins, self.fH, self.fW)

    def size_hint(self):
        """This plan's counts, or — while they are still on their way — those of the last plan of the same shape whose counts
        have arrived, 5 % up (a calibration that moved by a frame's ego motion keeps its counts to a fraction of a percent).
        Good for READ-AHEAD sizes only: nothing whose correctness depends on a count may use a hint."""
        c = self.counts()
        if c is not None:
            return c
        _poll_pending()
        h = _LAST_COUNTS.get(self._shape_key())
        if h is None:
            return None
        return dict(points=min(self.n_total, int(h["points"] * 1.05)), rows=min(self.n_rows, int(h["rows"] * 1.05)),
                    tiles_per_xcd=min(self.tiles_cap // 8, int(h["tiles_per_xcd"] * 1.05) + 1))

    def forward_tables(self):
        """The (approximately) valid prefixes of the forward tables, for read-ahead; None while nothing is known of their size."""
        c = self.size_hint()
        if c is None or c["points"] == 0:
            return None
        return [self.desc32[:8 * c["tiles_per_xcd"]], self.ivl_rel[:c["rows"]], self.pt[:c["points"]]]

    def reference_tables(self):
        """(ranks_bev in this plan's row numbering, ranks_depth, ranks_feat) as the reference orders them — needs
        ``keep_sorted=True`` at build time; blocks for the counts."""
        if self.rows_sorted is None:
            raise RuntimeError("build the plan with keep_sorted=True")
        n = self.n_points
        rd = self.ranks_depth_sorted[:n]
        return self.rows_sorted[:n], rd, ops.ranks_feat_from_depth(rd.contiguous(), self.depth_bins, self.feat_hw)


_LAST_COUNTS = {}       # shape key -> counts of the most recent plan of that shape whose counts reached the host
_PENDING = []           # weak references to plans whose counts are still in flight


def _poll_pending():
    """Collect the counts that have arrived since the last look (event queries, no waiting)."""
    alive = []
    for ref in _PENDING:
        p = ref()
        if p is not None and p.counts() is None:
            alive.append(ref)
    _PENDING[:] = alive[-8:]


def device_plan_supported(B, N, D, fH, fW, nx, channels=64):
    """The limits of the two kernels a device plan feeds (k_pool_fwd_direct / k_pool_bwd_patch, C = 64) and of the packed tables."""
    n_rows = B * int(nx[0]) * int(nx[1]) * int(nx[2])
    n_pix = B * N * fH * fW
    return (channels == 64 and 0 < D <= 127 and n_rows < 0xffffff and n_rows * 256 < 2 ** 32 and n_pix * 64 * 4 < 2 ** 31
            and n_pix * D * 4 < 2 ** 32 - 256 and n_pix * D < 0x3fffffff and 2 * D * 16 * 4 <= 64 * 1024)


def device_plans_enabled():
    """OMNIHD_POOL_DEVICE_PLAN=0 sends every new calibration through the host-scheduled ``plan.build_plan`` again."""
    return _env("OMNIHD_POOL_DEVICE_PLAN", "1") != "0"


BUILDS = {"device_plans": 0}      # how many plans this process built on the device (bench.py: fast_paths)


def build_device_plan(dx, bx, nx, layout="byxz", geom=None, rots=None, trans=None, axes=None, keep_sorted=False):
    """Enqueue the build of a plan on the current stream and return it — no synchronisation.

    Geometry: either ``geom`` (B,N,D,fH,fW,3) fp32, or ``rots`` (B,N,3,3) / ``trans`` (B,N,3) on the device + ``axes`` =
    (xs (fW), ys (fH), ds (D)) of the frustum (the kernel forms the points with the rounding steps of ``get_geometry``)."""
    if layout not in ("bzyx", "byxz"):
        raise ValueError(layout)
    if geom is not None:
        ops._want(geom, torch.float32, "geom")
        if geom.dim() != 6 or geom.size(-1) != 3:
            raise ValueError("geom must be (B,N,D,H,W,3)")
        B, N, D, fH, fW, _ = geom.shape
        dev = geom.device
        rots_ = trans_ = xs = ys = ds = None
    else:
        if rots is None or trans is None or axes is None:
            raise ValueError("rots, trans and the frustum axes are needed when no geometry tensor is given")
        B, N = trans.shape[:2]
        rots_ = ops._want(rots.reshape(B * N, 9).float().contiguous(), torch.float32, "rots")
        trans_ = ops._want(trans.reshape(B * N, 3).float().contiguous(), torch.float32, "trans")
        xs, ys, ds = (ops._want(a, torch.float32, "frustum axis") for a in axes)
        D, fH, fW = ds.numel(), ys.numel(), xs.numel()
        dev = trans_.device
    X, Y, Z = int(nx[0]), int(nx[1]), int(nx[2])
    if not device_plan_supported(B, N, D, fH, fW, (X, Y, Z)):
        raise ValueError("this frustum / grid does not fit the device-built plan (see device_plan_supported)")
    dxf = np.asarray(dx, dtype=np.float32)
    bxf = np.asarray(bx, dtype=np.float32)
    off = (bxf - dxf / np.float32(2.0)).astype(np.float32)          # the reference's fp32 tensor arithmetic (:328)
    plan = DevicePoolPlan(layout, (B, Z, Y, X), (B * N, D, fH, fW), dev)
    walk, band = _walk_on(dev, B * N, fH, fW)
    i32 = dict(dtype=torch.int32, device=dev)
    rows_sorted = torch.empty(plan.n_total, **i32)
    rd_sorted = torch.empty(plan.n_total, **i32)
    h_off = (ctypes.c_float * 3)(*off.tolist())
    h_dx = (ctypes.c_float * 3)(*dxf.tolist())
    h_nx = (ctypes.c_int * 3)(X, Y, Z)
    with ops._on(dev):
        ws = ops._workspace(plan.workspace_bytes, dev)
        _plan._timed("plan", lambda: check(lib().omnihd_pool_plan_build(
            _ptr(geom), _ptr(rots_), _ptr(trans_), _ptr(xs), _ptr(ys), _ptr(ds), B, N, D, fH, fW, ctypes.cast(h_off, ctypes.c_void_p),
            ctypes.cast(h_dx, ctypes.c_void_p), ctypes.cast(h_nx, ctypes.c_void_p), 1 if layout == "byxz" else 0, _ptr(walk),
            _ptr(band), _plan.TILE_ITEMS, _plan.LONG_LEN, _ptr(plan.pt), _ptr(plan.ivl_rel), _ptr(plan.desc32), _ptr(plan.row_ptr),
            _ptr(plan.row_bin), _ptr(plan.pix_ptr), _ptr(plan.patch_order), _ptr(plan.hdr), _ptr(rows_sorted), _ptr(rd_sorted),
            _ptr(ws), ws.numel(), ops._stream()), "omnihd_pool_plan_build"))
        # the counts follow the build to pinned host memory; nobody waits for them (see DevicePoolPlan.counts)
        plan._hdr_host = torch.empty(32, dtype=torch.int32, pin_memory=True)
        plan._hdr_host.copy_(plan.hdr, non_blocking=True)
        plan._hdr_event = torch.cuda.Event()
        plan._hdr_event.record()
    if keep_sorted:
        plan.rows_sorted, plan.ranks_depth_sorted = rows_sorted, rd_sorted
    _poll_pending()
    _PENDING.append(weakref.ref(plan))
    BUILDS["device_plans"] += 1
    return plan


# ---- output buffers whose empty rows are kept, across calibrations ------------------------------------------------------------
# omnihd_amd.plan keeps such buffers PER PLAN: which rows are empty is a property of the calibration.  With a calibration per
# frame the buffer outlives the plan; each buffer remembers the row CSR of the tables that filled it last, and the kernel
# zero-fills exactly the rows that were occupied then and are empty now (mode 2 of omnihd_bev_pool_v2_fwd_direct_dev).
_FAMILY = {}


class _FamilyKeeper(_plan._Keeper):
    __slots__ = ("row_ptr",)

    def __init__(self, tensor, base):
        super().__init__(tensor, base)
        self.row_ptr = None               # CSR of the plan that filled the buffer last; None: the buffer is all zeros


def _family_output(plan, c, device):
    """(keeper, empty_rows_mode, prev_row_ptr) or (None, 0, None).  Same guards as ``plan._kept_output``."""
    if not _plan._use_count_works():
        return None, 0, None
    kept = _FAMILY.setdefault((device.index, plan.n_rows, c), [])
    for k in kept:
        t = k.tensor
        if _plan._storage_users(t) == k.base:
            if t._version != k.version:
                _plan._warn_once(("inplace-family", plan.n_rows), "omnihd_amd: a pooled BEV tensor obtained with keep_empty_rows=True "
                                 "was written in place; its buffer is zero-filled again.  Callers that write into the result must "
                                 "not pass keep_empty_rows.")
                t.zero_()
                k.row_ptr = None
            k.version = t._version
            if k.row_ptr is None or k.row_ptr is plan.row_ptr:
                return k, 1, None
            return k, 2, k.row_ptr
    if len(kept) >= _plan.MAX_KEPT_OUTPUTS:
        return None, 0, None
    nbytes = plan.n_rows * c * 4
    if _plan._KEPT_TOTAL[0] + nbytes > int(_env("OMNIHD_POOL_KEEP_MAX_MB", "2048")) * (1 << 20):
        return None, 0, None
    buf = torch.zeros((plan.n_rows, c), dtype=torch.float32, device=device)
    base = _plan._storage_users(buf)
    if base is None:
        return None, 0, None
    _plan._KEPT_TOTAL[0] += nbytes
    k = _FamilyKeeper(buf, base)
    kept.append(k)
    return k, 1, None


def _forward_direct_dev(depth, feat, plan, out, mode, prev_row_ptr):
    n_feat_rows = feat.numel() // 64
    with ops._on(feat.device):
        check(lib().omnihd_bev_pool_v2_fwd_direct_dev(_ptr(depth), _ptr(feat), _ptr(plan.pt), _ptr(plan.ivl_rel), plan.ivl_rel.numel(),
                                                      _ptr(plan.desc32), _ptr(plan.hdr), plan.launch_slots(), _ptr(plan.row_ptr),
                                                      _ptr(prev_row_ptr), _ptr(out), 64, plan.n_rows, plan.depth_bins, plan.feat_hw,
                                                      n_feat_rows, mode, ops._stream()), "omnihd_bev_pool_v2_fwd_direct_dev")


def _check_matches(plan, depth, feat):
    if (depth.dim() != 5 or feat.dim() != 5 or feat.size(-1) != 64 or depth.size(0) * depth.size(1) != plan.n_img
            or depth.size(2) != plan.depth_bins or depth.size(3) != plan.fH or depth.size(4) != plan.fW
            or feat.numel() != plan.n_pix * 64):
        raise ValueError(f"depth {tuple(depth.shape)} / feat {tuple(feat.shape)} do not match the pooling plan "
                         f"(images {plan.n_img}, D {plan.depth_bins}, fH x fW {plan.fH} x {plan.fW}, C 64)")
    if feat.data_ptr() % 16:
        raise ValueError("feat must be 16-byte aligned")


class _DevicePlannedPool(torch.autograd.Function):
    """depth (B,N,D,H,W), feat (B,N,H,W,64) -> dense rows (n_rows, 64) in the plan's row order."""

    @staticmethod
    def forward(ctx, depth, feat, plan, keep_empty_rows=False):
        depth = depth.contiguous().float()
        feat = feat.contiguous().float()
        _check_matches(plan, depth, feat)
        keeper, mode, prev = _family_output(plan, 64, feat.device) if keep_empty_rows else (None, 0, None)
        fp = _plan.FAST_PATHS
        fp["pool_fwd_calls"] += 1
        fp["kept_output"] += keeper is not None
        fp["direct_fwd"] += 1
        if keeper is not None:
            out = keeper.tensor.view(plan.n_rows, 64)               # a VIEW: the guards of plan._kept_output apply
        else:
            out = torch.empty((plan.n_rows, 64), dtype=torch.float32, device=feat.device)
        _plan._timed("fwd", lambda: _forward_direct_dev(depth, feat, plan, out, mode, prev))
        if keeper is not None:
            keeper.row_ptr = plan.row_ptr
        ctx.save_for_backward(depth, feat)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, out_grad):
        depth, feat = ctx.saved_tensors
        plan = ctx.plan
        if out_grad.dtype != torch.float32 and _env("OMNIHD_POOL_PREFETCH", "1") != "0":
            c = plan.size_hint()
            ops.prefetch([plan.row_bin[:c["points"]] if c is not None else None, depth, feat])
        out_grad = out_grad.contiguous().float()
        ops.wgrad_overlap_fence(out_grad.device)
        try:
            depth_grad, feat_grad = torch.empty_like(depth), torch.empty_like(feat)      # both written densely
            _plan._timed("bwd", lambda: ops.bev_pool_v2_backward_patch(out_grad.view(plan.n_rows, 64), depth, feat, None,
                                                                       plan.row_bin, plan.pix_ptr, plan.patch_order, depth_grad,
                                                                       feat_grad))
            return depth_grad, feat_grad, None, None
        finally:
            if depth.is_cuda and not torch.is_grad_enabled():
                ops.wgrad_overlap_arm()


def device_planned_pool(depth, feat, plan, keep_empty_rows=False):
    """The pooled BEV tensor with logical shape (B, C, Z, Y, X) (ops/bev_pool_v2/bev_pool.py:86-92) — see ``plan.planned_pool``."""
    B, Z, Y, X = plan.grid
    rows = _DevicePlannedPool.apply(depth.float(), feat.float(), plan, bool(keep_empty_rows))
    if plan.layout == "bzyx":
        return rows.view(B, Z, Y, X, 64).permute(0, 4, 1, 2, 3)
    return rows.view(B, Y, X, Z, 64).permute(0, 4, 3, 1, 2)
