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
    pix_desc: torch.Tensor = None   # int32 [8*k, 4] schedule of the scheduled backward (every pixel once)
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


def pixel_schedule(bp_ranks_feat, bp_starts, bp_lengths, n_feat_rows, feat_hw=None, n_xcd=8):
    """Schedule of the scheduled backward: one descriptor {pixel row, first point, #points, 0} per
    image-feature pixel (pixels without points included, so feat_grad is written densely), walked
    in 4x4 pixel patches (neighbouring pixels hit the same BEV rows -> out_grad rows are reused from
    L1/L2) and cut into ``n_xcd`` contiguous runs, one per XCD.  Host-side planning, once per
    calibration; only the ORDER is a performance choice."""
    dev = bp_ranks_feat.device
    start = torch.zeros(n_feat_rows, dtype=torch.int32, device=dev)
    length = torch.zeros(n_feat_rows, dtype=torch.int32, device=dev)
    if bp_starts.numel():
        pix = bp_ranks_feat[bp_starts.long()].long()
        start[pix] = bp_starts
        length[pix] = bp_lengths
    f = torch.arange(n_feat_rows, device=dev)
    if feat_hw is not None:
        fH, fW = feat_hw
        img, h, w = f // (fH * fW), (f // fW) % fH, f % fW
        key = ((img * ((fH + 3) // 4) + h // 4) * ((fW + 3) // 4) + w // 4) * 16 + (h % 4) * 4 + (w % 4)
        f = f[torch.argsort(key, stable=True)]
    per = (n_feat_rows + n_xcd - 1) // n_xcd
    desc = torch.zeros((n_xcd * per, 4), dtype=torch.int32, device=dev)
    desc[:, 0] = -1
    # equal pixel counts per XCD; run k of the walk -> rows [k*per, k*per + count)
    bounds = [(n_feat_rows * k) // n_xcd for k in range(n_xcd + 1)]
    for k in range(n_xcd):
        run = f[bounds[k]:bounds[k + 1]]
        desc[k * per:k * per + run.numel(), 0] = run.int()
        desc[k * per:k * per + run.numel(), 1] = start[run]
        desc[k * per:k * per + run.numel(), 2] = length[run]
    return desc.contiguous()


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


def _patch_geometry(n_img, feat_hw, patch_w):
    fH, fW = feat_hw
    pw, ph = patch_w, PATCH // patch_w
    pcols, prows = (fW + pw - 1) // pw, (fH + ph - 1) // ph
    return pw, ph, pcols, prows, n_img * pcols * prows


def shared_schedule(n_img, feat_hw, patch_w, cost, n_xcd=8):
    """Order of the patches of the stream (shared-row) backward: image by image in bands of 4 image rows, left to right inside a band
    (vertically adjacent patches touch the same output rows: what one fetched is in L2 for the next), cut into ``n_xcd``
    contiguous runs of equal COST (``cost`` per patch, points + a fixed cost), heaviest bands first inside a run (see
    ``patch_schedule``).  Returns a list of ``n_xcd`` int64 tensors of patch ids; every patch appears exactly once."""
    pw, ph, pcols, prows, n_patch = _patch_geometry(n_img, feat_hw, patch_w)
    p = torch.arange(n_patch)
    img, pr, pc = p // (pcols * prows), (p // pcols) % prows, p % pcols
    band_rows = max(1, 4 // ph)
    nb = (prows + band_rows - 1) // band_rows
    band = img * nb + pr // band_rows
    col_block = (pc * pw) // PATCH                                  # 16 image columns at a time
    key = (band * ((pcols * pw + PATCH - 1) // PATCH + 1) + col_block) * (band_rows * (PATCH // pw)) + (pr % band_rows) * (PATCH // pw) + pc % (PATCH // pw)
    order = p[torch.argsort(key, stable=True)]
    c = cost[order].double()
    cum = torch.cumsum(c, 0)
    targets = cum[-1] * torch.arange(1, n_xcd, dtype=torch.float64) / n_xcd
    cuts = [0] + torch.searchsorted(cum, targets).tolist() + [n_patch]
    bands = band[order]
    runs = []
    for k_ in range(n_xcd):
        run, rb, rc = order[cuts[k_]:cuts[k_ + 1]], bands[cuts[k_]:cuts[k_ + 1]], c[cuts[k_]:cuts[k_ + 1]]
        if run.numel():
            ids, inv = torch.unique(rb, return_inverse=True)
            tot = torch.zeros(ids.numel(), dtype=torch.float64).index_add_(0, inv, rc)
            cnt = torch.zeros(ids.numel(), dtype=torch.float64).index_add_(0, inv, torch.ones_like(rc))
            rank = torch.argsort(torch.argsort(-(tot / cnt), stable=True), stable=True)       # heaviest band first
            run = run[torch.argsort(rank[inv], stable=True)]
        runs.append(run)
    return runs


def _patch_rows(bp_ranks_row, bp_ranks_depth, pix_ptr, n_img, depth_bins, feat_hw, patch_w, R):
    """What the shared-row (stream) backward needs of the backward tables (points sorted by pixel, inside a pixel by output row):
    per patch the sorted distinct rows (``uniq_rows`` with CSR ``u_ptr``), per point the stage-relative word, per patch and stage
    the 16 pixels' offsets into their point lists.  None when a pixel's list is not sorted by row."""
    dev = bp_ranks_row.device
    fH, fW = feat_hw
    fhw = fH * fW
    pw, ph, pcols, prows, n_patch = _patch_geometry(n_img, feat_hw, patch_w)
    n = int(bp_ranks_row.numel())
    pp = pix_ptr.long()
    lens = pp[1:] - pp[:-1]
    f = torch.repeat_interleave(torch.arange(n_img * fhw, device=dev), lens)            # pixel of every point
    rows = bp_ranks_row.long()
    if n > 1 and bool(((rows[1:] < rows[:-1]) & (f[1:] == f[:-1])).any()):
        return None                                                                     # a pixel's points must be sorted by row
    img, h, w = f // fhw, (f % fhw) // fW, f % fW
    patch = (img * prows + h // ph) * pcols + w // pw
    g = (h % ph) * pw + w % pw                                                          # the pixel's lane group inside its patch
    key = patch * (1 << 24) + rows
    uk, inv = torch.unique(key, return_inverse=True)                                    # sorted: by patch, then by row
    nu = torch.bincount(uk >> 24, minlength=n_patch)
    u_ptr = torch.zeros(n_patch + 1, dtype=torch.int64, device=dev)
    u_ptr[1:] = nu.cumsum(0)
    n_stage = (nu + R - 1) // R
    idx = inv - u_ptr[patch]                                                            # index of the point's row among its patch's rows
    stage = idx // R
    dbin = torch.div(bp_ranks_depth.long(), fhw, rounding_mode="floor") % depth_bins
    pt_word = (((idx % R) << 8) | (dbin << 24)).to(torch.int32).contiguous()
    rows_p = torch.clamp(n_stage, min=1) + 1                                            # offset rows of a patch: one per stage + the end (a patch without points: one empty stage)
    so_ptr = torch.zeros(n_patch + 1, dtype=torch.int64, device=dev)                    # first px_stage_off row of every patch
    so_ptr[1:] = rows_p.cumsum(0)
    total = int(so_ptr[-1].item())
    cnt = torch.zeros(total * PATCH, dtype=torch.int64, device=dev)
    if n:
        cnt.index_add_(0, (so_ptr[patch] + stage + 1) * PATCH + g, torch.ones(n, dtype=torch.int64, device=dev))
    cs = cnt.view(total, PATCH).cumsum(0)                                               # points of lane group g in all stages before this row
    first = torch.repeat_interleave(so_ptr[:-1], rows_p)
    rel = cs - cs[first]                                                                # [total, 16] offsets inside the pixel's list
    pts = torch.bincount(patch, minlength=n_patch) if n else torch.zeros(n_patch, dtype=torch.int64, device=dev)
    # first point (index into the backward tables) of pixel g of every patch; pixels outside the image: 0 (they have no points)
    pid = torch.arange(n_patch, device=dev)
    p_img, p_r, p_c = pid // (pcols * prows), (pid // pcols) % prows, pid % pcols
    gg = torch.arange(PATCH, device=dev)
    hh, ww = p_r[:, None] * ph + gg[None, :] // pw, p_c[:, None] * pw + gg[None, :] % pw
    inside = (hh < fH) & (ww < fW)
    fpix = (p_img[:, None] * fhw + hh * fW + ww).clamp(max=n_img * fhw - 1)
    px_start = torch.where(inside, pp[fpix], torch.zeros_like(fpix))
    return dict(n_patch=n_patch, uniq=(uk & 0xffffff).to(torch.int32).contiguous(), u_ptr=u_ptr, nu=nu, n_stage=n_stage,
                pt_word=pt_word, so_ptr=so_ptr, rel=rel, pts=pts, px_start=px_start, reuse=float(n) / max(1, int(uk.numel())))


STREAM_FIRST, STREAM_LAST, STREAM_VALID = 1 << 30, 1 << 29, 1 << 28     # flags of a stream entry (csrc/bev_pool_v2.hip: kStream*)
STREAM_STAGE_COST = 48        # cost model of the dealing: a stage costs as much as this many points, a patch as PATCH_FIXED_COST/4


@dataclass
class StreamBackwardTables:
    """Tables of the stream backward (include/omnihd_hip.h: omnihd_bev_pool_v2_bwd_stream)."""
    patch_w: int
    rows_per_stage: int
    pt_word: torch.Tensor       # int32 [Npts]: 256 * (index of the point's row inside its stage) | depth bin << 24
    uniq_rows: torch.Tensor     # int32: per patch, the sorted distinct output rows its points touch
    px_off: torch.Tensor        # int32 [(rows + 1) * 16]: per patch and stage, the index into pt_word of pixel g's first point of the stage
    stream: torch.Tensor        # int32 [n_entries, 4]
    stream_ptr: torch.Tensor    # int32 [n_streams + 1] entries of wave w: [ptr[w], ptr[w+1])
    n_streams: int
    reuse: float
    balance: float              # heaviest stream / mean stream (cost model)


def stream_deal(runs, cost, streams_per_xcd):
    """Deal the patches of every XCD run (in walk order) to ``streams_per_xcd`` waves: the next patch goes to the wave with the
    least cost so far, so that the waves of an XCD move through the walk together and finish together.  Returns one list of
    patch ids per stream (XCD-major)."""
    import heapq
    out = []
    c = cost.tolist()
    for run in runs:
        heap = [(0.0, w) for w in range(streams_per_xcd)]
        lists = [[] for _ in range(streams_per_xcd)]
        for pid in run.tolist():
            load, w = heapq.heappop(heap)
            lists[w].append(pid)
            heapq.heappush(heap, (load + c[pid], w))
        out += lists
    return out


def stream_tables_from(bp_ranks_row, bp_ranks_depth, pix_ptr, n_img, depth_bins, feat_hw, patch_w=8, rows_per_stage=64,
                       streams_per_xcd=192):
    """Tables of the stream backward: the stages of ``_patch_rows`` laid out as one stream per wave (see the kernel's header
    comment for the entry format).  Pure torch + a host heap for the dealing; None when the tables do not fit the kernel."""
    if patch_w not in (16, 8, 4) or rows_per_stage not in (32, 48, 64) or not 0 < depth_bins <= 64:
        return None
    R = rows_per_stage
    t = _patch_rows(bp_ranks_row, bp_ranks_depth, pix_ptr, n_img, depth_bins, feat_hw, patch_w, R)
    if t is None or t["n_patch"] >= (1 << 28) or int(t["so_ptr"][-1]) >= (1 << 24):
        return None                                             # the entry fields: 28 bits of patch id, 24 bits of offset row
    dev = bp_ranks_row.device
    n_stage = t["n_stage"].cpu()
    n_ent = torch.clamp(n_stage, min=1)                         # a patch without points still has one (empty) stage: its gradients are zeros
    cost = (t["pts"].cpu() + STREAM_STAGE_COST * n_ent + PATCH_FIXED_COST // 4).double()
    runs = shared_schedule(n_img, feat_hw, patch_w, cost)
    lists = stream_deal(runs, cost, streams_per_xcd)
    n_streams = len(lists)
    order = torch.tensor([pid for l in lists for pid in l], dtype=torch.int64)          # patches in stream order
    per_stream = torch.tensor([len(l) for l in lists], dtype=torch.int64)
    sid_of_patch = torch.repeat_interleave(torch.arange(n_streams), per_stream)
    # global stage list G: stream-major, inside a stream patch by patch, stage by stage
    ne = n_ent[order]
    g_patch = torch.repeat_interleave(order, ne)
    g_sid = torch.repeat_interleave(sid_of_patch, ne)
    g_first_idx = torch.zeros(order.numel() + 1, dtype=torch.int64)
    g_first_idx[1:] = ne.cumsum(0)
    g_k = torch.arange(int(g_first_idx[-1])) - torch.repeat_interleave(g_first_idx[:-1], ne)
    nu, u_ptr, so_ptr = t["nu"].cpu(), t["u_ptr"].cpu(), t["so_ptr"].cpu()
    g_first = g_k == 0
    g_last = g_k == ne[torch.repeat_interleave(torch.arange(order.numel()), ne)] - 1
    g_ustart = u_ptr[g_patch] + g_k * R
    g_nrows = torch.clamp(nu[g_patch] - g_k * R, min=0, max=R)
    g_so = so_ptr[g_patch] + g_k
    n_g = g_patch.numel()
    stages_per_stream = torch.bincount(g_sid, minlength=n_streams)
    gs = torch.zeros(n_streams + 1, dtype=torch.int64)
    gs[1:] = stages_per_stream.cumsum(0)
    es = gs + 2 * torch.arange(n_streams + 1)                   # entries: two more than stages per stream
    n_entries = int(es[-1])
    e_sid = torch.repeat_interleave(torch.arange(n_streams), stages_per_stream + 2)
    e_j = torch.arange(n_entries) - es[e_sid]
    ent = torch.zeros((n_entries, 4), dtype=torch.int64)
    ent[:, 3] = -1
    # x: the stage consumed in this iteration (entry j consumes stage j - 2 of its stream)
    has = e_j >= 2
    gi = (gs[e_sid] + e_j - 2)[has]
    ent[has, 0] = g_patch[gi] | STREAM_VALID | torch.where(g_first[gi], STREAM_FIRST, 0) | torch.where(g_last[gi], STREAM_LAST, 0)
    # y, z: the stage whose row ids / offsets are requested (stage j)
    has = e_j < stages_per_stream[e_sid]
    gi = (gs[e_sid] + e_j)[has]
    ent[has, 1] = g_ustart[gi]
    ent[has, 2] = g_so[gi] | (g_nrows[gi] << 24)
    # w: the patch whose first stage is stage j - 1 (its pixel data are requested one iteration ahead)
    has = (e_j >= 1) & (e_j <= stages_per_stream[e_sid])
    gi = (gs[e_sid] + e_j - 1)[has]
    ent[has, 3] = torch.where(g_first[gi], g_patch[gi], torch.full_like(gi, -1))
    # absolute offsets into pt_word (+ one row of padding: the kernel reads the row behind a patch's last one)
    rel, px_start = t["rel"], t["px_start"]
    patch_of_row = torch.repeat_interleave(torch.arange(t["n_patch"], device=dev), torch.clamp(t["n_stage"], min=1) + 1)
    px_off = rel + px_start[patch_of_row]
    px_off = torch.cat([px_off, px_off.new_zeros(1, PATCH)]).to(torch.int32).contiguous().view(-1)
    load = torch.zeros(n_streams, dtype=torch.float64).index_add_(0, sid_of_patch, cost[order])
    return StreamBackwardTables(patch_w, R, t["pt_word"], t["uniq"], px_off, ent.to(torch.int32).contiguous().to(dev),
                                es.to(torch.int32).contiguous().to(dev), n_streams, t["reuse"],
                                float(load.max() / load.mean().clamp(min=1e-9)))


STREAM_DEFAULT = (8, 32, 256)   # (patch width, rows per stage, waves per XCD) of the stream backward unless OMNIHD_POOL_BWD_STREAM_SHAPE="w,R,waves"


def stream_tables(plan, n_img, feat_hw2):
    """The stream backward's tables of ``plan``, built on first use and kept with it (None: does not fit, see above)."""
    shape = _env("OMNIHD_POOL_BWD_STREAM_SHAPE", "")
    pw, R, spx = (int(v) for v in shape.split(",")) if shape else STREAM_DEFAULT
    cache = plan.__dict__.setdefault("_stream", {})
    key = (pw, R, spx, n_img, tuple(feat_hw2))
    if key not in cache:
        cache[key] = stream_tables_from(plan.bp_ranks_row, plan.bp_ranks_depth, plan.pix_ptr, n_img, plan.depth_bins, feat_hw2, pw, R, spx)
    return cache[key]


def _finish(layout, grid, rows, rd, rf, starts, lengths, n_feat_rows, feat_hw=None, origin_cell=None):
    B, Z, Y, X = grid
    n_rows = B * Z * Y * X
    row_ptr = ops.csr_from_sorted_keys(rows, n_rows)
    tile_row = ops.csr_tiles(row_ptr, TILE_ITEMS, LONG_LEN)
    tile_order = tile_schedule(row_ptr, tile_row, rf, feat_hw, grid=grid, layout=layout, origin_cell=origin_cell)
    tile_desc = ops.tile_descriptors(row_ptr, tile_row, tile_order)
    bp = ops.backward_tables(rows, rd, rf, n_feat_rows)
    pix_desc = pixel_schedule(bp[2], bp[3], bp[4], n_feat_rows, feat_hw)
    plan = BevPoolPlan(layout, grid, n_rows, int(rows.numel()), rows, rd, rf, row_ptr, tile_row, tile_order,
                       tile_desc, starts, lengths, bp[0], bp[1], bp[2], bp[3], bp[4], pix_desc)
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


def _lean_forward():
    """The one-table forward kernel (k_pool_fwd_lean) is the default when the plan knows its frustum geometry;
    OMNIHD_POOL_LEAN=0 selects the three-table kernel (bit-identical results, 43.2 vs 51.2 us at R1 in one run)."""
    import os
    return _env("OMNIHD_POOL_LEAN", "1") != "0"


def _direct_forward():
    """The direct forward (k_pool_fwd_direct, C = 64: lane groups walk their piece of the point list from global memory, no LDS
    record staging) is the default where it applies; OMNIHD_POOL_DIRECT=0 selects k_pool_fwd_lean2 (same tiles; rows cut by
    the in-tile split may differ in the last bit)."""
    return _env("OMNIHD_POOL_DIRECT", "1") != "0"


def forward_tables(plan, channels=64):
    """The static tables the forward kernel of ``plan`` will read for ``channels`` feature channels (what the LSS module streams
    into the caches ahead of the launch, ``ops.prefetch``): the direct kernel's point words, row ids and 32-int descriptors, or
    the lean kernels' tile descriptors, CSR and rank table."""
    if hasattr(plan, "launch_slots"):                     # pool_plan.DevicePoolPlan: valid prefixes once its counts are known
        return plan.forward_tables() or []
    if channels == 64 and plan.depth_bins > 0 and _lean_forward() and _direct_forward():
        pt, ivl_rel, desc32 = direct_tables(plan)
        return [desc32, ivl_rel, pt]
    return [plan.tile_desc, plan.row_ptr, plan.ranks_depth]


def _patch_backward():
    """The patch backward (k_pool_bwd_patch, C = 64) is the default; OMNIHD_POOL_BWD_PATCH=0 selects the scheduled kernel."""
    import os
    return _env("OMNIHD_POOL_BWD_PATCH", "1") != "0"


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
    """depth (B,N,D,H,W), feat (B,N,H,W,C) -> dense rows (n_rows, C) in the plan's row order."""

    @staticmethod
    def forward(ctx, depth, feat, plan, keep_empty_rows=False):
        depth = depth.contiguous().float()
        feat = feat.contiguous().float()
        _check_plan_matches(plan, depth, feat)
        lean = plan.depth_bins > 0 and plan.tile_desc is not None and plan.n_points > 0 and _lean_forward()
        # the limits of omnihd_bev_pool_v2_fwd_direct (csrc/bev_pool_v2.hip): C = 64, 32-bit gather offsets into feat / depth
        direct = (lean and feat.size(-1) == 64 and feat.numel() * 4 < 2 ** 31 and depth.numel() * 4 < 2 ** 32 - 8
                  and depth.numel() < 0x3fffffff and feat.data_ptr() % 16 == 0 and _direct_forward())
        # only the direct and the second-generation lean kernel know how to leave the empty rows alone
        can_keep = direct or (lean and _env("OMNIHD_POOL_LEAN2", "1") != "0" and feat.numel() * 4 < 2 ** 31)
        keeper = _kept_output(plan, feat.size(-1), feat.device) if (keep_empty_rows and can_keep) else None
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
        elif lean:
            _timed("fwd", lambda: ops.bev_pool_v2_forward_lean(depth, feat, plan.ranks_depth, plan.row_ptr, plan.tile_desc, out,
                                                               plan.depth_bins, plan.feat_hw, empty_rows_kept=keeper is not None))
        else:
            ops.bev_pool_v2_forward_csr(depth, feat, plan.ranks_depth, plan.ranks_feat, plan.row_ptr, out,
                                        plan.ranks_row, plan.tile_desc)
        ctx.save_for_backward(depth, feat)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, out_grad):
        depth, feat = ctx.saved_tensors
        plan = ctx.plan
        c = feat.size(-1)
        # the limits of omnihd_bev_pool_v2_bwd_patch (csrc/bev_pool_v2.hip): C = 64, 32-bit gather offsets into out_grad and a
        # 24-bit row field, both D x 16 LDS blocks within 64 KiB, 16-byte aligned row tensors; anything else takes the
        # scheduled kernel
        patch = (c == 64 and plan.patch_order is not None and plan.depth_bins > 0 and depth.dim() == 5
                 and plan.n_rows * 256 < 2 ** 32 and plan.n_rows < 0xffffff and plan.depth_bins <= 512
                 and feat.data_ptr() % 16 == 0 and _patch_backward())
        if patch and out_grad.dtype != torch.float32 and _env("OMNIHD_POOL_PREFETCH", "1") != "0":
            # the forward's tensors and the backward tables have long left the Infinity Cache: read them ahead on the side
            # stream while the cast of the incoming gradient runs (bf16 step; in the fp32 step nothing precedes the kernel)
            packed = _row_bin(plan)
            ops.prefetch([packed, depth, feat] if packed is not None else [plan.bp_ranks_depth, plan.bp_ranks_row, depth, feat])
        out_grad = out_grad.contiguous().float()
        ops.wgrad_overlap_fence(out_grad.device)       # (a no-op unless OMNIHD_WGRAD_OVERLAP=all put weight gradients in flight)
        try:
            return _PlannedPool._backward(ctx, out_grad, depth, feat, plan, c, patch)
        finally:
            if depth.is_cuda and not torch.is_grad_enabled():
                ops.wgrad_overlap_arm()                # the convolutions behind this point overlap their weight gradients

    @staticmethod
    def _backward(ctx, out_grad, depth, feat, plan, c, patch):
        if patch and _env("OMNIHD_POOL_BWD_STREAM", "0") == "1" and plan.depth_bins <= 64 and depth.numel() * 4 < 2 ** 32 - 256:
            # opt-in: the stream form of the same arithmetic (rows of a patch gathered once; DESIGN 4.2: not faster, so not the default)
            st = stream_tables(plan, depth.size(0) * depth.size(1), (depth.size(3), depth.size(4)))
            if st is not None:
                depth_grad, feat_grad = torch.empty_like(depth), torch.empty_like(feat)
                _timed("bwd", lambda: ops.bev_pool_v2_backward_stream(out_grad.view(plan.n_rows, c), depth, feat, st, depth_grad, feat_grad))
                return depth_grad, feat_grad, None, None
        if patch:
            depth_grad, feat_grad = torch.empty_like(depth), torch.empty_like(feat)   # both written densely
            packed = _row_bin(plan)
            _timed("bwd", lambda: ops.bev_pool_v2_backward_patch(out_grad.view(plan.n_rows, c), depth, feat,
                                                                 None if packed is not None else plan.bp_ranks_depth,
                                                                 packed if packed is not None else plan.bp_ranks_row, plan.pix_ptr,
                                                                 plan.patch_order, depth_grad, feat_grad))
            return depth_grad, feat_grad, None, None
        depth_grad = torch.zeros_like(depth)
        if plan.pix_desc is not None and c in (4, 8, 16, 32, 64):
            feat_grad = torch.empty_like(feat)          # written densely by the scheduled kernel
            ops.bev_pool_v2_backward_sched(out_grad.view(plan.n_rows, c), depth, feat, plan.bp_ranks_depth,
                                           plan.bp_ranks_row, plan.pix_desc, depth_grad, feat_grad)
        else:
            feat_grad = torch.zeros_like(feat)
            og5 = out_grad.view(1, 1, 1, plan.n_rows, c)
            ops.bev_pool_v2_backward(og5, depth_grad, feat_grad, depth, feat, plan.bp_ranks_depth,
                                     plan.bp_ranks_feat, plan.bp_ranks_row, plan.bp_lengths, plan.bp_starts)
        return depth_grad, feat_grad, None, None


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
