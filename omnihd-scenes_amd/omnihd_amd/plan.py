"""Cached pooling plans.

The five rank tables of the reference are a pure function of the camera calibration and the
grid, yet the reference rebuilds them in every forward
(cam_stream_lss_bevpoolv2_depthnet.py:281-283) and re-sorts all points in every backward
(ops/bev_pool_v2/bev_pool.py:47-57).  A ``BevPoolPlan`` holds, on the device,

* the forward tables grouped by OUTPUT ROW in CSR form (``row_ptr``), so the dense forward kernel
  writes the whole BEV tensor once (no zero-fill, no permute copy, no s2c concat copy), and
* the backward tables grouped by image-feature pixel,

for one of two row numberings: ``'bzyx'`` (the reference's (B,Z,Y,X,C) buffer) or ``'byxz'``
((B,Y,X,Z,C) memory = the channels-last layout of the s2c tensor (B, Z*C, Y, X)).
"""
import os
from ._env import env as _env
from dataclasses import dataclass

import torch

from . import ops


@dataclass
class BevPoolPlan:
    layout: str            # 'bzyx' | 'byxz'
    grid: tuple            # (B, Z, Y, X)
    n_rows: int            # B*Z*Y*X
    n_points: int
    # forward (sorted by row)
    ranks_row: torch.Tensor     # int32 [n_points]  row index of each point (sorted ascending)
    ranks_depth: torch.Tensor   # int32 [n_points]
    ranks_feat: torch.Tensor    # int32 [n_points]
    row_ptr: torch.Tensor       # int32 [n_rows+1]
    tile_row: torch.Tensor      # int32 [n_tiles+1] work partition of the tiled forward kernel
    tile_order: torch.Tensor    # int32 [8*ceil(n_tiles/8)] launch schedule (XCD x slot -> tile, -1 idle)
    tile_desc: torch.Tensor     # int32 [8*ceil(n_tiles/8), 4] the schedule as kernel descriptors
    interval_starts: torch.Tensor
    interval_lengths: torch.Tensor
    # backward (sorted by ranks_feat)
    bp_ranks_row: torch.Tensor
    bp_ranks_depth: torch.Tensor
    bp_ranks_feat: torch.Tensor
    bp_starts: torch.Tensor
    bp_lengths: torch.Tensor
    pix_ptr: torch.Tensor = None    # int32 [n_feat_rows+1] CSR of the backward tables over image-feature pixels
    patch_order: torch.Tensor = None  # int32 [8*k] schedule of the patch backward (16-pixel patches, -1 idle)
    bp_row_bin: torch.Tensor = None   # int32 [Npts] (output row | depth bin << 24) in backward order: the patch backward's one table
    depth_bins: int = 0             # D and fH*fW of the frustum the plan was built from (0: unknown, e.g. foreign
    feat_hw: int = 0                # tables): with them the forward derives ranks_feat from ranks_depth in-kernel

    @property
    def n_intervals(self):
        return int(self.interval_starts.numel())


TILE_ITEMS = 768     # rows+points per tile of the tiled forward
LONG_LEN = 512       # a row with more points is a tile of its own (TILE_ITEMS + LONG_LEN <= 1280)


def tile_schedule(row_ptr, tile_row, ranks_feat, feat_hw=None, n_xcd=8, grid=None, layout="byxz", origin_cell=None):
    """Launch schedule of the tiled forward: which tile each (XCD, slot) works on.

    Host-side planning only (runs once per calibration).  The workgroups resident on one XCD (block
    b -> XCD b % 8) should gather from as few image-feature rows as possible so that they stay in that
    XCD's 4 MiB L2.  With ``grid`` (B,Z,Y,X) the tiles are ordered by the AZIMUTH of their BEV cells
    around ``origin_cell`` (x, y in cells; default grid centre — the LSS module passes the centroid of
    the camera positions) and cut into ``n_xcd`` runs of equal work: a wedge of BEV is seen through
    a narrow band of image columns.  Measured on the R1 rig: 2.5-3.9 MB of feature rows per XCD
    (24 MB summed, 17.3 MB compulsory) against 3.8-6.8 MB (45 MB) for an ordering by mean image column
    and 53 MB for contiguous BEV bands.  Without ``grid`` the mean image column is used.
    Changes speed only: every tile is processed exactly once whatever the order."""
    n_tiles = tile_row.numel() - 1
    per = (n_tiles + n_xcd - 1) // n_xcd          # the kernel derives the same value from n_tiles
    dev = row_ptr.device
    lo = row_ptr[tile_row[:-1].long()].long()
    hi = row_ptr[tile_row[1:].long()].long()
    cnt = hi - lo
    work = (cnt + (tile_row[1:] - tile_row[:-1]).long()).double()
    if grid is not None:
        B, Z, Y, X = grid
        mid = ((tile_row[:-1] + tile_row[1:]) // 2).long().clamp(max=B * Z * Y * X - 1)
        if layout == "byxz":
            cell = mid // Z
            yy, xx = (cell // X) % Y, cell % X
        else:
            yy, xx = (mid // X) % Y, mid % X
        ox, oy = origin_cell if origin_cell is not None else ((X - 1) / 2.0, (Y - 1) / 2.0)
        key = torch.atan2(yy.double() - oy, xx.double() - ox)
        order = torch.argsort(key, stable=True)
    else:
        if feat_hw is not None:
            fH, fW = feat_hw
            col = (ranks_feat // (fH * fW)) * fW + ranks_feat % fW
        else:
            col = ranks_feat
        csum = torch.cat([torch.zeros(1, dtype=torch.float64, device=dev), col.double().cumsum(0)])
        mean = (csum[hi] - csum[lo]) / cnt.clamp(min=1)
        mean = torch.where(cnt > 0, mean, torch.full_like(mean, float("inf")))
        order = torch.argsort(mean, stable=True)
    # cut the ordered tiles into n_xcd runs of (nearly) equal work, at most ``per`` tiles each
    cw = work[order].cumsum(0)
    run = torch.clamp((cw * n_xcd / (cw[-1] + 1)).long(), max=n_xcd - 1)
    flat = torch.full((n_xcd * per,), -1, dtype=torch.int64, device=dev)
    spill = []
    for k in range(n_xcd):
        mine = order[run == k]
        if mine.numel() > per:
            spill.append(mine[per:])
            mine = mine[:per]
        flat[k * per:k * per + mine.numel()] = mine
    if spill:
        extra = torch.cat(spill)
        free = torch.nonzero(flat < 0).flatten()[:extra.numel()]
        flat[free] = extra
    assert int((flat >= 0).sum()) == n_tiles
    return flat.int().contiguous()


PATCH = 16   # pixels per patch of the patch backward (one 64-byte segment of depth / depth_grad per depth bin)


PATCH_FIXED_COST = 400        # a patch's set-up + epilogue in units of points (9 us vs 45 points/us, pool_bwd_trace.py)


def patch_schedule(n_img, feat_hw, n_xcd=8, pix_ptr=None):
    """Launch schedule of the patch backward: patch p = pixels [16*(p % ppi), 16*(p % ppi) + 16) of image p // ppi,
    ppi = ceil(fH*fW / 16).  The patches are walked image by image in blocks of 4 image rows (vertically adjacent pixels
    see the same BEV cells at neighbouring heights, so the out_grad rows one patch fetched are still in L2 for the next)
    and cut into ``n_xcd`` contiguous runs, one per XCD.  With ``pix_ptr`` (points per pixel) the cut equalises the runs'
    COST (points + a fixed cost per patch) instead of their length, and inside a run the 4-row blocks are issued heaviest
    first: the workgroups that start last are the cheap ones (the static equal-count schedule ran 2048 workgroups until
    36 us of a 58 us launch and then drained for 22 us).  int32 [n_xcd * per], -1 = idle slot.  Host-side planning, once
    per calibration; only the ORDER is a performance choice, every patch appears exactly once."""
    fH, fW = feat_hw
    fhw = fH * fW
    ppi = (fhw + PATCH - 1) // PATCH
    p = torch.arange(n_img * ppi)
    img, k = p // ppi, p % ppi
    h, w = (k * PATCH) // fW, (k * PATCH) % fW
    key = ((img * ((fH + 3) // 4) + h // 4) * ((fW + PATCH - 1) // PATCH + 1) + w // PATCH) * 4 + h % 4
    order = p[torch.argsort(key, stable=True)]
    n = order.numel()
    if pix_ptr is None:
        bounds = [(n * k) // n_xcd for k in range(n_xcd + 1)]
        runs = [order[bounds[k]:bounds[k + 1]] for k in range(n_xcd)]
    else:
        pp = pix_ptr.detach().cpu().long()
        lens = (pp[1:] - pp[:-1]).view(n_img, fhw)
        pad = ppi * PATCH - fhw
        if pad:
            lens = torch.cat([lens, lens.new_zeros(n_img, pad)], dim=1)
        cost = lens.view(n_img, ppi, PATCH).sum(-1).view(-1) + PATCH_FIXED_COST          # per patch, indexed by p
        c = cost[order].double()
        cum = torch.cumsum(c, 0)
        targets = cum[-1] * torch.arange(1, n_xcd, dtype=torch.float64) / n_xcd
        cuts = [0] + torch.searchsorted(cum, targets).tolist() + [n]
        block = (img * ((fH + 3) // 4) + h // 4)[order]                                  # 4-row block of every entry
        runs = []
        for k_ in range(n_xcd):
            run, rb, rc = order[cuts[k_]:cuts[k_ + 1]], block[cuts[k_]:cuts[k_ + 1]], c[cuts[k_]:cuts[k_ + 1]]
            if run.numel():
                ids, inv = torch.unique(rb, return_inverse=True)
                tot = torch.zeros(ids.numel(), dtype=torch.float64).index_add_(0, inv, rc)
                cnt = torch.zeros(ids.numel(), dtype=torch.float64).index_add_(0, inv, torch.ones_like(rc))
                rank = torch.argsort(torch.argsort(-(tot / cnt), stable=True), stable=True)   # heaviest block first
                run = run[torch.argsort(rank[inv], stable=True)]
            runs.append(run)
    per = max(1, max(r.numel() for r in runs))
    flat = torch.full((n_xcd * per,), -1, dtype=torch.int32)
    for k_ in range(n_xcd):
        flat[k_ * per:k_ * per + runs[k_].numel()] = runs[k_].int()
    return flat.contiguous()


DIRECT_GROUPS = 16   # lane groups per workgroup of k_pool_fwd_direct (C = 64: 16 lanes per output row)


def direct_tables_from(ranks_row, ranks_depth, tile_row, tile_desc):
    """Tables of the direct forward kernel (include/omnihd_hip.h: omnihd_bev_pool_v2_fwd_direct) from the plan's forward
    tables; pure torch (host-side planning, once per calibration; works on CPU tensors for the tests).

    pt[p]      = ranks_depth[p] | closing[p] << 31, closing = p is the last point of its output row;
    ivl_rel[k] = output row of the k-th non-empty row, relative to the first row of the tile that holds it;
    desc32[s]  = {first row, #rows, first point, #points, 0,0,0,0, g[16], 0 x 8}: group j of the workgroup walks points
                 [first + min(j*w, n), first + min((j+1)*w, n)), w = ceil(n/16); g[j] = number of non-empty rows closed before
                 its first point | 1 << 31 when that point continues the row of the point in front of it (same tile)."""
    dev = ranks_row.device
    rows = ranks_row.long()
    n = rows.numel()
    S = tile_desc.size(0)
    desc32 = torch.zeros((S, 32), dtype=torch.int64, device=dev)
    desc32[:, :4] = tile_desc.long()
    if n == 0:
        return (torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev),
                desc32.int().contiguous())
    closing = torch.ones(n, dtype=torch.bool, device=dev)
    closing[:-1] = rows[1:] != rows[:-1]
    rd = ranks_depth.long()
    pt = torch.where(closing, rd - (1 << 31), rd).to(torch.int32).contiguous()          # rd | 1 << 31 as a signed word
    closed_before = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    closed_before[1:] = closing.long().cumsum(0)
    crow = rows[closing]
    tr = tile_row.long()
    ivl_rel = (crow - tr[torch.searchsorted(tr, crow, right=True) - 1]).to(torch.int32).contiguous()
    Pa, npts = desc32[:, 2], desc32[:, 3]
    w = (npts + DIRECT_GROUPS - 1) // DIRECT_GROUPS
    g = torch.arange(DIRECT_GROUPS, device=dev)
    off = torch.minimum(g[None, :] * w[:, None], npts[:, None])
    q = Pa[:, None] + off                                                               # first point of every group's piece
    inside = off < npts[:, None]
    head = inside & (off > 0) & ~closing[(q - 1).clamp(min=0, max=n - 1)]
    gi = closed_before[q.clamp(max=n)] + head.long() * (1 << 31)
    gi = torch.where(gi >= (1 << 31), gi - (1 << 32), gi)
    desc32[:, 8:8 + DIRECT_GROUPS] = torch.where(inside, gi, torch.zeros_like(gi))
    return pt, ivl_rel, desc32.to(torch.int32).contiguous()


def direct_tables(plan):
    """(pt, ivl_rel, desc32) of ``plan`` for the direct forward kernel, built on first use and kept with the plan."""
    got = getattr(plan, "_direct", None)
    if got is None:
        got = plan._direct = direct_tables_from(plan.ranks_row, plan.ranks_depth, plan.tile_row, plan.tile_desc)
    return got


def _finish(layout, grid, rows, rd, rf, starts, lengths, n_feat_rows, feat_hw=None, origin_cell=None):
    B, Z, Y, X = grid
    n_rows = B * Z * Y * X
    row_ptr = ops.csr_from_sorted_keys(rows, n_rows)
    tile_row = ops.csr_tiles(row_ptr, TILE_ITEMS, LONG_LEN)
    tile_order = tile_schedule(row_ptr, tile_row, rf, feat_hw, grid=grid, layout=layout, origin_cell=origin_cell)
    tile_desc = ops.tile_descriptors(row_ptr, tile_row, tile_order)
    bp = ops.backward_tables(rows, rd, rf, n_feat_rows)
    plan = BevPoolPlan(layout, grid, n_rows, int(rows.numel()), rows, rd, rf, row_ptr, tile_row, tile_order,
                       tile_desc, starts, lengths, bp[0], bp[1], bp[2], bp[3], bp[4])
    if feat_hw is not None and n_feat_rows % (feat_hw[0] * feat_hw[1]) == 0:
        plan.pix_ptr = ops.csr_from_sorted_keys(bp[2], n_feat_rows)
        plan.patch_order = patch_schedule(n_feat_rows // (feat_hw[0] * feat_hw[1]), feat_hw, pix_ptr=plan.pix_ptr).to(rows.device)
    return plan


def build_plan(coor, dx, bx, nx, layout="byxz", origin_xy=None):
    """Plan from frustum geometry (B,N,D,H,W,3) — one fused key pass + two radix sorts.
    ``origin_xy``: metric (x, y) of the rig centre (centroid of the camera positions) for the XCD schedule."""
    if layout not in ("bzyx", "byxz"):
        raise ValueError(layout)
    B, N, D, H, W, _ = coor.shape
    X, Y, Z = int(nx[0]), int(nx[1]), int(nx[2])
    keys, idx, sentinel = ops.rank_keys(coor.contiguous(), dx, bx, nx)
    if layout == "byxz":
        keys = ops.permute_rows_zyx_to_yxz(keys, Z, Y, X)   # the sentinel maps onto itself
    rows, (rd,), starts, lengths = ops.sort_ranks(keys, [idx], ops._bits_for(sentinel), sentinel)
    rows, rd = rows.contiguous(), rd.contiguous()
    rf = ops.ranks_feat_from_depth(rd, D, H * W)
    origin_cell = None
    if origin_xy is not None:
        origin_cell = ((float(origin_xy[0]) - (float(bx[0]) - float(dx[0]) / 2)) / float(dx[0]) - 0.5,
                       (float(origin_xy[1]) - (float(bx[1]) - float(dx[1]) / 2)) / float(dx[1]) - 0.5)
    plan = _finish(layout, (B, Z, Y, X), rows, rd, rf, starts.contiguous(), lengths.contiguous(), B * N * H * W,
                   feat_hw=(H, W), origin_cell=origin_cell)
    plan.depth_bins, plan.feat_hw = int(D), int(H * W)
    return plan


def plan_from_tables(ranks_bev, ranks_depth, ranks_feat, grid, n_feat_rows, layout="bzyx", feat_hw=None):
    """Plan from reference-format tables (sorted by ranks_bev in (b,z,y,x) numbering)."""
    B, Z, Y, X = grid
    if layout == "bzyx":
        rows, (rd, rf), starts, lengths = ops.sort_ranks(
            ranks_bev.contiguous(), [ranks_depth.contiguous(), ranks_feat.contiguous()],
            ops._bits_for(B * Z * Y * X))
    else:
        keys = ops.permute_rows_zyx_to_yxz(ranks_bev.contiguous(), Z, Y, X)
        rows, (rd, rf), starts, lengths = ops.sort_ranks(
            keys, [ranks_depth.contiguous(), ranks_feat.contiguous()], ops._bits_for(B * Z * Y * X))
    return _finish(layout, grid, rows.contiguous(), rd.contiguous(), rf.contiguous(), starts.contiguous(),
                   lengths.contiguous(), n_feat_rows, feat_hw=feat_hw)


def _direct_ok(plan, channels):
    """The product's forward kernel (k_pool_fwd_direct) takes C = 64 and a plan that knows its frustum geometry."""
    return channels == 64 and plan.depth_bins > 0 and plan.tile_desc is not None


def forward_tables(plan, channels=64):
    """The static tables the forward kernel of ``plan`` will read for ``channels`` feature channels (what the LSS module streams
    into the caches ahead of the launch, ``ops.prefetch``): the direct kernel's descriptors, row ids and point words, or the
    CSR and rank tables of the any-channel kernel."""
    if hasattr(plan, "launch_slots"):                     # pool_plan.DevicePoolPlan: valid prefixes once its counts are known
        return plan.forward_tables() or []
    if _direct_ok(plan, channels):
        pt, ivl_rel, desc32 = direct_tables(plan)
        return [desc32, ivl_rel, pt]
    return [plan.row_ptr, plan.ranks_depth, plan.ranks_feat]


# How often the optional fast paths of the pooling forward were actually taken in this process (bench.py reports them as
# `fast_paths`: two of them rest on private torch hooks that are probed and may silently be off after a torch upgrade)
FAST_PATHS = {"pool_fwd_calls": 0, "kept_output": 0, "direct_fwd": 0}

# bench.py sets this to a list to collect (start, end) event pairs around every pooling forward / backward kernel launched
# inside the training step (the in-step launch duration the roofline line is computed from); None = no events
TIMING = None


def _timed(kind, launch):
    if TIMING is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    TIMING.append((kind, e0, e1))


def _check_plan_matches(plan, depth, feat):
    """A plan built from one frustum must not be run on tensors of another shape: the kernels index depth / feat by the
    plan's tables (an out-of-range rank is a wild device read, not an exception)."""
    if plan.depth_bins <= 0:
        return                                       # foreign tables (plan_from_tables): the caller vouches for them
    c = feat.size(-1)
    n_feat_rows = feat.numel() // max(c, 1)
    if depth.dim() != 5 or depth.size(2) != plan.depth_bins or depth.size(3) * depth.size(4) != plan.feat_hw:
        raise ValueError(f"depth {tuple(depth.shape)} does not match the pooling plan (D={plan.depth_bins}, fH*fW={plan.feat_hw})")
    if depth.numel() != n_feat_rows * plan.depth_bins:
        raise ValueError(f"depth {tuple(depth.shape)} and feat {tuple(feat.shape)} disagree on the number of image pixels")
    if plan.pix_ptr is not None and plan.pix_ptr.numel() != n_feat_rows + 1:
        raise ValueError(f"feat {tuple(feat.shape)} has {n_feat_rows} pixel rows, the plan was built for {plan.pix_ptr.numel() - 1}")


def _storage_users(t):
    """How many owners the tensor's storage has right now (tensors, views, saved tensors of autograd nodes); None when this
    torch build does not expose the count."""
    try:
        return torch._C._storage_Use_Count(t.untyped_storage()._cdata)
    except (AttributeError, RuntimeError, TypeError):
        return None


_USE_COUNT_OK = None


def _use_count_works():
    """``torch._C._storage_Use_Count`` is a private hook: trust it only after it has counted a view coming and going on a
    scratch tensor in THIS process (absent, or with another meaning in a later torch: the kept-buffer path is simply off)."""
    global _USE_COUNT_OK
    if _USE_COUNT_OK is None:
        t = torch.zeros(4)
        n0 = _storage_users(t)
        v = t.view(2, 2)
        n1 = _storage_users(t)
        del v
        n2 = _storage_users(t)
        _USE_COUNT_OK = n0 is not None and n1 == n0 + 1 and n2 == n0
    return _USE_COUNT_OK


MAX_KEPT_OUTPUTS = 2      # output buffers kept per plan (one in flight between forward and backward + one spare)
_KEPT_TOTAL = [0]         # bytes of all live kept buffers of this process (every plan), bounded by OMNIHD_POOL_KEEP_MAX_MB
_WARNED = set()


def _warn_once(key, msg):
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


class _Keeper:
    """One kept output buffer of a plan: the tensor, the owner count of its storage when nobody else holds it, and the tensor's
    version counter after our last launch (kernels write through raw pointers and do not move it; every in-place torch
    operation on the buffer OR on a view of it does)."""
    __slots__ = ("tensor", "base", "version", "__weakref__")

    def __init__(self, tensor, base):
        self.tensor, self.base, self.version = tensor, base, tensor._version

    def __del__(self):
        try:
            _KEPT_TOTAL[0] -= self.tensor.numel() * 4
        except Exception:
            pass


def _empty_row_index(plan):
    idx = getattr(plan, "_empty_rows", None)
    if idx is None:
        idx = plan._empty_rows = torch.nonzero(plan.row_ptr[1:] == plan.row_ptr[:-1]).flatten()
    return idx


def _kept_output(plan, c, device):
    """An output buffer of this plan whose EMPTY rows are zero already, or None.

    40 % of the BEV rows of a camera rig collect no frustum point at all (248 596 of 614 400 at R1) — which ones is a property
    of the plan, i.e. of the calibration.  The dense forward has to leave zeros there; the reference zero-fills the whole
    tensor every forward (ops/bev_pool_v2/bev_pool.py:27).  A buffer that a previous forward of the SAME plan produced
    still holds those zeros as long as nobody wrote to it, so the kernel stores only the rows that collect points (94 MB
    instead of 157 MB per launch at R1).  Guards:
      * a buffer is handed out again only when every other owner of its storage is gone (the result tensor, its views, the
        copy the next layer saved for its backward): while a result is alive its memory is never touched;
      * the result is handed out as a VIEW of the kept tensor, so (i) under autograd torch itself refuses in-place writes into
        it ("a view created inside a custom Function ... is being modified inplace"), and (ii) without autograd an in-place
        write moves the kept tensor's version counter: the buffer is then zero-filled again before its next use (one extra
        fill, result still right) and a warning names the cause;
      * OMNIHD_POOL_VERIFY_ZEROS=1 additionally sums the empty rows before every reuse (a host synchronisation: debugging
        aid against writers that bypass torch, e.g. foreign kernels on raw pointers);
      * the private use-count hook is probed once per process (``_use_count_works``), and the bytes kept by all plans are
        bounded (OMNIHD_POOL_KEEP_MAX_MB, default 2048): beyond that, or with the hook missing, the plain path runs."""
    if not _use_count_works():
        return None
    kept = getattr(plan, "_kept_outputs", None)
    if kept is None:
        kept = plan._kept_outputs = []
    for k in kept:
        t = k.tensor
        if t.shape[1] == c and t.device == device and _storage_users(t) == k.base:
            if t._version != k.version:
                _warn_once(("inplace", id(plan)), "omnihd_amd: a pooled BEV tensor obtained with keep_empty_rows=True was written in "
                           "place; its buffer is zero-filled again (results stay right, the saving of the kept rows is lost for "
                           "this step).  Callers that write into the result must not pass keep_empty_rows.")
                t.zero_()
            elif _env("OMNIHD_POOL_VERIFY_ZEROS", "0") == "1":
                idx = _empty_row_index(plan)
                if idx.numel() and float(t.index_select(0, idx).abs().sum()) != 0.0:
                    _warn_once(("dirty", id(plan)), "omnihd_amd: OMNIHD_POOL_VERIFY_ZEROS found non-zero values in rows no frustum "
                               "point reaches (somebody wrote into a kept pooling buffer behind torch's back); zero-filled again")
                    t.zero_()
            k.version = t._version
            return k
    if len(kept) >= MAX_KEPT_OUTPUTS:
        return None
    nbytes = plan.n_rows * c * 4
    if _KEPT_TOTAL[0] + nbytes > int(_env("OMNIHD_POOL_KEEP_MAX_MB", "2048")) * (1 << 20):
        return None
    keeper = torch.zeros((plan.n_rows, c), dtype=torch.float32, device=device)          # zero-filled once
    base = _storage_users(keeper)
    if base is None:
        return None
    _KEPT_TOTAL[0] += nbytes
    k = _Keeper(keeper, base)
    kept.append(k)
    return k


def _row_bin(plan):
    """(output row | depth bin << 24) per point in backward order, built once per plan on the device: the patch backward then
    reads ONE table word per point (the bin was ``(ranks_depth // (fH*fW)) % D``).  None when the fields do not fit or
    OMNIHD_POOL_BWD_PACKED=0."""
    if plan.bp_row_bin is None and _env("OMNIHD_POOL_BWD_PACKED", "1") != "0":
        if 0 < plan.depth_bins <= 127 and plan.n_rows < 0xffffff and plan.feat_hw > 0:
            d = torch.div(plan.bp_ranks_depth, plan.feat_hw, rounding_mode="floor") % plan.depth_bins
            plan.bp_row_bin = (plan.bp_ranks_row | (d << 24)).to(torch.int32).contiguous()
    return plan.bp_row_bin if _env("OMNIHD_POOL_BWD_PACKED", "1") != "0" else None


class _PlannedPool(torch.autograd.Function):
    """depth (B,N,D,H,W), feat (B,N,H,W,C) -> dense rows (n_rows, C) in the plan's row order.

    C = 64 (every configuration of the reference): k_pool_fwd_direct / k_pool_bwd_patch.  Any other channel count: the
    any-channel dense kernel forward and the reference-format backward kernel on the plan's backward tables."""

    @staticmethod
    def forward(ctx, depth, feat, plan, keep_empty_rows=False):
        depth = depth.contiguous().float()
        feat = feat.contiguous().float()
        _check_plan_matches(plan, depth, feat)
        # the limits of omnihd_bev_pool_v2_fwd_direct (csrc/bev_pool_v2.hip): C = 64, 32-bit gather offsets into feat / depth
        direct = (_direct_ok(plan, feat.size(-1)) and plan.n_points > 0 and feat.numel() * 4 < 2 ** 31
                  and depth.numel() * 4 < 2 ** 32 - 8 and depth.numel() < 0x3fffffff and feat.data_ptr() % 16 == 0)
        keeper = _kept_output(plan, feat.size(-1), feat.device) if (keep_empty_rows and direct) else None
        FAST_PATHS["pool_fwd_calls"] += 1
        FAST_PATHS["kept_output"] += keeper is not None
        FAST_PATHS["direct_fwd"] += bool(direct)
        if keeper is not None:
            out = keeper.tensor.view(plan.n_rows, feat.size(-1))    # a VIEW: see the guards listed in _kept_output
        else:
            out = torch.empty((plan.n_rows, feat.size(-1)), dtype=torch.float32, device=feat.device)
        if direct:
            pt, ivl_rel, desc32 = direct_tables(plan)
            _timed("fwd", lambda: ops.bev_pool_v2_forward_direct(depth, feat, pt, ivl_rel, desc32, plan.row_ptr, out, plan.depth_bins,
                                                                 plan.feat_hw, empty_rows_kept=keeper is not None))
        else:
            ops.bev_pool_v2_forward_csr(depth, feat, plan.ranks_depth, plan.ranks_feat, plan.row_ptr, out)
        ctx.save_for_backward(depth, feat)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, out_grad):
        depth, feat = ctx.saved_tensors
        plan = ctx.plan
        c = feat.size(-1)
        # the limits of omnihd_bev_pool_v2_bwd_patch (csrc/bev_pool_v2.hip): C = 64, 32-bit gather offsets into out_grad and a
        # 24-bit row field, both D x 16 LDS blocks within 64 KiB, 16-byte aligned row tensors
        patch = (c == 64 and plan.patch_order is not None and plan.depth_bins > 0 and depth.dim() == 5
                 and plan.n_rows * 256 < 2 ** 32 and plan.n_rows < 0xffffff and plan.depth_bins <= 512
                 and feat.data_ptr() % 16 == 0)
        if patch and out_grad.dtype != torch.float32 and _env("OMNIHD_POOL_PREFETCH", "1") != "0":
            # the forward's tensors and the backward tables have long left the Infinity Cache: read them ahead on the side
            # stream while the cast of the incoming gradient runs (bf16 step; in the fp32 step nothing precedes the kernel)
            packed = _row_bin(plan)
            ops.prefetch([packed, depth, feat] if packed is not None else [plan.bp_ranks_depth, plan.bp_ranks_row, depth, feat])
        out_grad = out_grad.contiguous().float()
        ops.wgrad_overlap_fence(out_grad.device)       # (a no-op unless OMNIHD_WGRAD_OVERLAP=all put weight gradients in flight)
        try:
            if patch:
                depth_grad, feat_grad = torch.empty_like(depth), torch.empty_like(feat)   # both written densely
                packed = _row_bin(plan)
                _timed("bwd", lambda: ops.bev_pool_v2_backward_patch(out_grad.view(plan.n_rows, c), depth, feat,
                                                                     None if packed is not None else plan.bp_ranks_depth,
                                                                     packed if packed is not None else plan.bp_ranks_row, plan.pix_ptr,
                                                                     plan.patch_order, depth_grad, feat_grad))
            else:
                depth_grad, feat_grad = torch.zeros_like(depth), torch.zeros_like(feat)
                og5 = out_grad.view(1, 1, 1, plan.n_rows, c)
                ops.bev_pool_v2_backward(og5, depth_grad, feat_grad, depth, feat, plan.bp_ranks_depth, plan.bp_ranks_feat,
                                         plan.bp_ranks_row, plan.bp_lengths, plan.bp_starts)
            return depth_grad, feat_grad, None, None
        finally:
            if depth.is_cuda and not torch.is_grad_enabled():
                ops.wgrad_overlap_arm()                # the convolutions behind this point overlap their weight gradients


def planned_pool(depth, feat, plan, keep_empty_rows=False):
    """Returns the pooled BEV tensor with logical shape (B, C, Z, Y, X) (what the reference's
    ``bev_pool_v2`` returns, ops/bev_pool_v2/bev_pool.py:86-92).  For ``layout='byxz'`` it is a
    zero-copy view over (B,Y,X,Z,C) memory, so ``cat(unbind(dim=2), 1)`` (s2c) is a reshape.
    ``keep_empty_rows``: reuse an output buffer of the same plan whose empty rows are zero already (see ``_kept_output``);
    only for callers that never write into the result in place."""
    if hasattr(plan, "launch_slots"):                     # a plan built on the device (omnihd_amd/pool_plan.py)
        from .pool_plan import device_planned_pool
        return device_planned_pool(depth, feat, plan, keep_empty_rows)
    B, Z, Y, X = plan.grid
    C = feat.size(-1)
    # fp32 like the reference (bev_pool.py:20-21); the casts are autograd ops so bf16 callers get bf16 grads
    rows = _PlannedPool.apply(depth.float(), feat.float(), plan, bool(keep_empty_rows))
    if plan.layout == "bzyx":
        return rows.view(B, Z, Y, X, C).permute(0, 4, 1, 2, 3)
    return rows.view(B, Y, X, Z, C).permute(0, 4, 3, 1, 2)
