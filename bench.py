#!/usr/bin/env python3
"""bench.py — frames/sec of the camera + 4D-radar BEV-fusion hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic input that is already resident
in HBM.  Workloads (`--workload`):
  fusion    (default) one TRAINING step of the reference's camera + 4D-radar BEV-fusion detector
            (config projects/configs/bevfusion_NewScenes/bevfusion.py, BEVFUSION_depth: ResNet-50 +
            FPNC, DepthNet, LSS pooling, BEV encoder, radar voxelise + pillars + SECOND/FPN, fusion
            conv + SE, Anchor3DHead losses, KL depth loss): forward + backward + grad-clip + AdamW,
            at the BASELINE resolution R1 = 6 x (3 x 256 x 704) images + one merged (N x 7) radar
            cloud per frame (`--res r2` = the repo's 544 x 960 / 8 radar channels).  Timed twice by
            default: in fp32, the reference's arithmetic (`value`, `dtype` "f32"), and with the dense
            convolutions under bf16 autocast (`bf16_autocast`); pooling / voxelisation / losses are fp32
            (hand-written HIP) in both.
  bev_ops   only the north_star operators (pooling fwd+bwd, voxelise, scatter), no conv layers.
Besides the whole-job rate the line carries
  roofline      achieved HBM GB/s of the dominant kernel (bev_pool_v2 forward, dense), computed
                from ALGORITHMIC bytes / mean duration of the kernel's launches INSIDE
                the timed steps (HIP events on the launching stream); the isolated loops (back-to-back on
                rotating buffer sets = Infinity-Cache-warm reads; after a 512 MiB sweep = cold) beside it;
  cpu_baseline  the CPU oracle (a port of the reference algorithm; the reference has no CPU
                kernels) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "omnihd-scenes_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)



def _seed_miopen_db():
    """MIOpen tuning records for this workload's convolution geometries (omnihd_amd.harness.seed_miopen_db): offered to MIOpen as
    its user database before the first convolution runs, so that the find step of a fresh box is a lookup."""
    from omnihd_amd.harness import seed_miopen_db
    seed_miopen_db()


_seed_miopen_db()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md; measured 2495 with 32x32x16)
CONV_GEOMETRY = (1024, 1024, 3, 160, 240)     # (cin, cout, k, H, W) of the BEV encoder's first convolution: the largest dense layer
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); measured here: copy 5.2 TB/s, fill 6.5 TB/s (scripts/write_bw.py)
RES = {"r1": (256, 704, 410.0), "r2": (544, 960, 560.0)}
PC_RANGE = [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="fusion", choices=["fusion", "bev_ops"])
    ap.add_argument("--dtype", default="both", choices=["both", "bf16", "fp32"],
                    help="fusion workload: 'fp32' = the reference's arithmetic (it trains in fp32), 'bf16' = dense convolutions "
                         "under bf16 autocast; 'both' (default) times K steps of each, `value` is the fp32 rate and the bf16 "
                         "rate is carried in `bf16_autocast`")
    ap.add_argument("--res", default="r1", choices=list(RES))
    ap.add_argument("--batch", type=int, default=1, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-launches", type=int, default=60)
    ap.add_argument("--selftest-launch", action="store_true",
                    help="exercise only the rank launch / rendezvous / reporting skeleton (gloo, no GPU work): not a measurement")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# synthetic inputs; geometry through the same torch ops the reference uses
# ---------------------------------------------------------------------------------------------
def rig(res, batch, dev):
    """rots (B,6,3,3), trans (B,6,3) exactly as bevf_faster_rcnn_bevdepth.py:121-130 builds them
    (fp32 inverse of lidar2img on the host, then to the device)."""
    H, W, fx = RES[res]
    mats = []
    for yaw_deg in (0, 60, -60, 180, 120, -120):
        yaw = np.radians(yaw_deg)
        R_c2l = np.array([[np.sin(yaw), 0, np.cos(yaw)], [-np.cos(yaw), 0, np.sin(yaw)], [0, -1, 0]])
        t_c2l = np.array([np.cos(yaw), np.sin(yaw), 1.5])
        R = R_c2l.T
        E = np.eye(4); E[:3, :3] = R; E[:3, 3] = -R @ t_c2l
        K = np.eye(4); K[0, 0] = K[1, 1] = fx; K[0, 2] = W / 2; K[1, 2] = H / 2
        mats.append(K @ E)
    inv = [torch.Tensor(m).inverse() for m in mats]
    rots = torch.stack([m[:3, :3] for m in inv]).to(dev)[None].repeat(batch, 1, 1, 1)
    trans = torch.stack([m[:3, 3] for m in inv]).to(dev)[None].repeat(batch, 1, 1)
    return rots, trans


def lss_constants(res):
    """dx, bx, nx and frustum as the reference computes them (gen_dx_bx :80-85, create_frustum :222-233)."""
    H, W, _ = RES[res]
    rows = [[PC_RANGE[0], PC_RANGE[3], 0.5], [PC_RANGE[1], PC_RANGE[4], 0.5], [PC_RANGE[2], PC_RANGE[5], 0.5]]
    dx = torch.Tensor([r[2] for r in rows]).numpy()
    bx = torch.Tensor([r[0] + r[2] / 2.0 for r in rows]).numpy()
    nx = torch.LongTensor([(r[1] - r[0]) / r[2] for r in rows]).numpy()
    fH, fW = H // 4, W // 4
    ds = torch.arange(1, 60, 1, dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
    D = ds.shape[0]
    xs = torch.linspace(0, W - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
    ys = torch.linspace(0, H - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
    return dx, bx, nx, torch.stack((xs, ys, ds), -1)


def geometry(frustum, rots, trans):
    """get_geometry (:235-264) on the device, as the product computes it (LiftSplatShoot_Depth.get_geometry): three
    broadcast multiply-adds per axis in the reference's accumulation order — bit for bit the reference's torch-CPU
    result, without its one-3x3-GEMM-per-frustum-point batched matmul."""
    B, N, _ = trans.shape
    fr = frustum.to(rots.device)
    pz = fr[..., 2]
    px, py = fr[..., 0] * pz, fr[..., 1] * pz
    R, t = rots.view(B, N, 1, 1, 1, 3, 3), trans.view(B, N, 1, 1, 1, 3)
    axes = []
    for a in range(3):
        acc = R[..., a, 0] * px + R[..., a, 1] * py
        acc = acc + R[..., a, 2] * pz
        axes.append(acc + t[..., a])
    return torch.stack(axes, dim=-1)


def radar_points(rng, n):
    pts = np.empty((n, 7), dtype=np.float32)
    pts[:, 0] = rng.uniform(-60, 60, n); pts[:, 1] = rng.uniform(-40, 40, n); pts[:, 2] = rng.uniform(-3, 5, n)
    pts[:, 3:5] = rng.normal(0, 5, (n, 2)); pts[:, 5] = rng.uniform(0, 60, n); pts[:, 6] = rng.uniform(0, 40, n)
    return pts


class BevOps:
    """The north_star operators on one batch, with `sets` rotating input/output buffer sets."""

    def __init__(self, res, batch, dev, seed, sets=4):
        import omnihd_amd
        from omnihd_amd import ops
        omnihd_amd.require_gpu()
        self.ops, self.omnihd = ops, omnihd_amd
        self.dev, self.batch = dev, batch
        self.keep_empty = os.environ.get("OMNIHD_POOL_KEEP_ZEROS", "1") != "0"
        H, W, _ = RES[res]
        self.fH, self.fW, self.D, self.C, self.N = H // 4, W // 4, 59, 64, 6
        dx, bx, nx, frustum = lss_constants(res)
        self.nx = nx
        rots, trans = rig(res, batch, dev)
        geom = geometry(frustum, rots, trans).contiguous()
        self.plan = omnihd_amd.build_plan(geom, dx, bx, nx, layout="byxz")
        from omnihd_amd.plan import _row_bin, direct_tables
        self.row_bin = _row_bin(self.plan)
        self.direct_tabs = direct_tables(self.plan)
        del geom
        g = torch.Generator(device=dev).manual_seed(seed)
        self.sets = []
        for _ in range(sets):
            depth = torch.rand(batch, self.N, self.D, self.fH, self.fW, device=dev, generator=g).softmax(2)
            feat = torch.randn(batch, self.N, self.fH, self.fW, self.C, device=dev, generator=g)
            og = torch.randn(self.plan.n_rows, self.C, device=dev, generator=g)
            out = torch.zeros(self.plan.n_rows, self.C, device=dev)
            # private copies of the tables too, so that nothing is served from the Infinity Cache:
            # tb = (row_ptr, pix_ptr, patch_order, packed backward table, (pt, ivl_rel, desc32))
            tabs = [self.plan.row_ptr.clone(), self.plan.pix_ptr.clone(), self.plan.patch_order.clone(), self.row_bin.clone(),
                    [t.clone() for t in self.direct_tabs]]
            self.sets.append((depth, feat, og, out, torch.empty_like(depth), torch.empty_like(feat), tabs))
        rng = np.random.default_rng(seed)
        self.points = [torch.from_numpy(radar_points(rng, int(rng.integers(8000, 20001)))).to(dev) for _ in range(batch)]
        self.i = 0

    def pool_fwd(self, s):
        depth, feat, og, out, dg, fg, tb = self.sets[s]
        # as the product launches it: the empty rows of `out` are zero already (same plan, nobody wrote to it) and are kept
        pt, ivl_rel, desc32 = tb[4]
        self.ops.bev_pool_v2_forward_direct(depth, feat, pt, ivl_rel, desc32, tb[0], out, self.D, self.fH * self.fW,
                                            empty_rows_kept=self.keep_empty)

    def pool_bwd(self, s):
        depth, feat, og, out, dg, fg, tb = self.sets[s]
        # one packed table (row | depth bin << 24), as the plan's autograd function does
        self.ops.bev_pool_v2_backward_patch(og, depth, feat, None, tb[3], tb[1], tb[2], dg, fg)

    def radar(self):
        vox, coors, nums = [], [], []
        for b, pts in enumerate(self.points):
            v, c, n = self.ops.hard_voxelize(pts, [0.25, 0.25, 8], PC_RANGE, 10, 30000)
            vox.append(v); nums.append(n)
            coors.append(torch.nn.functional.pad(c, (1, 0), value=b))
        v, c = torch.cat(vox), torch.cat(coors)
        # stand-in pillar feature (the PFN is dense torch work, not part of bev_ops): mean of xyz.. padded to 64
        feats = v.sum(1)[:, :1].expand(-1, 64).contiguous()
        return self.ops.pillar_scatter(feats, c, self.batch, 320, 480)

    def other_ops(self, res, launches=20):
        """SURVEY.md 8(d) "Metric": achieved HBM GB/s of rank preparation and pillar scatter by their algorithmic bytes, and
        the duration + launch count of the (latency-bound, <= 5 MB) hard voxelisation.  Isolated launches, events on the
        launching stream; rank preparation contains one host read-back of two counts (wall time between synchronisations)."""
        dx, bx, nx, frustum = lss_constants(res)
        rots, trans = rig(res, self.batch, self.dev)
        geom = geometry(frustum, rots, trans).contiguous()
        ntot = geom.numel() // 3
        prep = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tabs = self.ops.voxel_pooling_prepare_v2(geom, dx, bx, nx)
            torch.cuda.synchronize()
            prep.append(time.perf_counter() - t0)
        npts, nint = tabs[0].numel(), tabs[3].numel()
        prep_bytes = 12 * ntot + 12 * npts + 8 * nint
        del geom, tabs
        pend = self.ops.hard_voxelize_async(self.points[0], [0.25, 0.25, 8], PC_RANGE, 10, 30000)
        v, c, n = pend.get()
        m = int(v.shape[0])
        feats = torch.randn(m, 64, device=self.dev)
        coors = torch.nn.functional.pad(c, (1, 0), value=0)

        def ev_time(fn):
            """Mean GPU time per call: the calls are enqueued while a spin kernel keeps the device busy, so the events bracket
            back-to-back execution and not the host's enqueue rate (these wrappers cost ~15 us of Python per call)."""
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(8_000_000)
            e0.record()
            for _ in range(launches):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / launches

        t_sc = ev_time(lambda: self.ops.pillar_scatter(feats, coors, 1, 320, 480, channels_last=True))
        t_vx = ev_time(lambda: self.ops.hard_voxelize_async(self.points[0], [0.25, 0.25, 8], PC_RANGE, 10, 30000))
        sc_bytes = 4 * 64 * 320 * 480 + 4 * 64 * m
        vx_bytes = 4 * self.points[0].shape[1] * self.points[0].shape[0] + 4 * (10 * self.points[0].shape[1] + 1 + 4) * m
        t_prep = sorted(prep)[len(prep) // 2]
        gbs = lambda b, t: round(b / t / 1e9, 1)
        # the whole plan of a NEW calibration on the device (csrc/pool_plan.hip): what a plan-cache miss costs inside a step
        from omnihd_amd import pool_plan
        H, W, _ = RES[res]
        xs = torch.linspace(0, W - 1, self.fW, dtype=torch.float, device=self.dev)
        ys = torch.linspace(0, H - 1, self.fH, dtype=torch.float, device=self.dev)
        ds = torch.arange(1, 60, 1, dtype=torch.float, device=self.dev)
        rigs = [tuple(x.contiguous() for x in (rots + 1e-4 * k, trans + 0.05 * k)) for k in range(4)]
        k_rig = [0]

        def build():
            r, tr = rigs[k_rig[0] % 4]
            k_rig[0] += 1
            return pool_plan.build_device_plan(dx, bx, nx, rots=r, trans=tr, axes=(xs, ys, ds))

        t_plan = ev_time(build)
        t0 = time.perf_counter()
        for _ in range(10):
            build()
        t_plan_host = (time.perf_counter() - t0) / 10
        torch.cuda.synchronize()
        plan_entry = {"algorithmic_bytes": prep_bytes, "mean_us": round(t_plan * 1e6, 1), "host_enqueue_us": round(t_plan_host * 1e6, 1),
                      "achieved_GBps": gbs(prep_bytes, t_plan), "frac": round(prep_bytes / t_plan / 1e9 / HBM_PEAK_GBS, 4),
                      "note": "EVERYTHING a new calibration needs, built on the device by one library call without a host read-back "
                              "(fused geometry + keys, radix sort, row CSR + scan, tiles + XCD schedule, direct-forward tables, "
                              "backward tables without a second sort, patch schedule: ~25 launches); device time of back-to-back "
                              "builds; algorithmic bytes = the SURVEY 8(d) rank-prep formula (the sort passes and the schedule "
                              "tables are overhead on top: a chain of small launches, latency- not bandwidth-bound)"}
        return {"plan_build": plan_entry,
                "rank_prep": {"algorithmic_bytes": prep_bytes, "median_us": round(t_prep * 1e6, 1), "achieved_GBps": gbs(prep_bytes, t_prep),
                              "frac": round(prep_bytes / t_prep / 1e9 / HBM_PEAK_GBS, 4),
                              "note": "reference-format five tables from (B,N,D,H,W,3) geometry: key pass + radix sort + RLE; wall time incl. one "
                                      "host read-back; runs once per calibration (plan cache), not per step"},
                "pillar_scatter": {"algorithmic_bytes": sc_bytes, "mean_us": round(t_sc * 1e6, 2), "achieved_GBps": gbs(sc_bytes, t_sc),
                                   "frac": round(sc_bytes / t_sc / 1e9 / HBM_PEAK_GBS, 4), "pillars": m,
                                   "note": "cell-map kernel + one dense pass that also resets the map (two launches); device time of back-to-back calls"},
                "hard_voxelize": {"algorithmic_bytes": vx_bytes, "mean_us": round(t_vx * 1e6, 1), "launches": 3, "points": int(self.points[0].shape[0]),
                                  "voxels": m, "note": "latency-bound (<= 5 MB): cell-grid voxeliser, 3 kernel launches + async count read-back per call (csrc/voxelize.hip, grid path)"}}

    def step(self):
        s = self.i % len(self.sets)
        self.i += 1
        self.pool_fwd(s)
        self.pool_bwd(s)
        self.radar()

    def fwd_kernel_name(self):
        return "k_pool_fwd_direct"

    def bwd_kernel_name(self):
        return "k_pool_bwd_patch"

    def bwd_algorithmic_bytes(self):
        """3 per-point tables + backward interval tables + depth gather + feature rows + touched out_grad rows
        + dense depth_grad + feat_grad."""
        npts, nint = self.plan.n_points, self.plan.n_intervals
        npix = self.batch * self.N * self.fH * self.fW
        ntot = npix * self.D
        nint_bp = int(self.plan.bp_starts.numel())
        return 4 * 3 * npts + 4 * 2 * nint_bp + 4 * npts + 4 * self.C * npix + 4 * self.C * nint + 4 * ntot + 4 * self.C * npix

    # ---- algorithmic bytes of the dense forward kernel, SURVEY.md 8(d) formula ----------------
    def fwd_algorithmic_bytes(self):
        npts, nint = self.plan.n_points, self.plan.n_intervals   # (the kernel reads row_ptr+ranks_row instead of
        # the 3 interval tables; the SURVEY formula is kept as the algorithmic figure)
        npix = self.batch * self.N * self.fH * self.fW
        nvox = self.plan.n_rows
        return 4 * (2 * npts) + 4 * (3 * nint) + 4 * npts + 4 * self.C * npix + 4 * self.C * nvox


def time_kernel(fn, n_sets, launches):
    """Mean duration of `launches` back-to-back launches (rotating buffer sets), HIP events on the
    launching stream (= torch's current stream)."""
    for k in range(max(4 * launches, 200)):          # bring the clocks up: a fresh process starts in a low power state
        fn(k % n_sets)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(launches):
        fn(k % n_sets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / launches


def time_kernel_cold(fn, n_sets, launches=12):
    """Mean duration of single launches each preceded by a 512 MiB read sweep (inputs AND tables cold in HBM; the
    back-to-back loop of `time_kernel` keeps the ~50 MB a launch reads resident in the 256 MiB Infinity Cache even with four
    rotating buffer sets: nt stores do not allocate there, scripts/lab/pool_context.py)."""
    sweep = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda")
    total = 0.0
    for k in range(launches):
        sweep.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(k % n_sets)
        e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1)
    return total * 1e-3 / launches


def cpu_baseline(res, budget_s=20.0):
    """Oracle (port of the reference algorithm) on the host cores: rank tables as the reference
    builds them every forward + pooling fwd + re-sort + pooling bwd + voxelise + scatter."""
    from oracle import cpu as OC
    from oracle import lss_oracle as O
    H, W, fx = RES[res]
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    dx, bx, nx = O.gen_dx_bx([PC_RANGE[0], PC_RANGE[3], 0.5], [PC_RANGE[1], PC_RANGE[4], 0.5], [PC_RANGE[2], PC_RANGE[5], 0.5])
    fr = O.create_frustum((H, W), 4, [1, 60, 1])
    inv = [torch.Tensor(m).inverse() for m in O.synthetic_rig(H, W, fx)]
    rots = torch.stack([m[:3, :3] for m in inv])[None].numpy()
    trans = torch.stack([m[:3, 3] for m in inv])[None].numpy()
    rng = np.random.default_rng(0)
    D, fH, fW, C = 59, H // 4, W // 4, 64
    depth = rng.random((1, 6, D, fH, fW), dtype=np.float32)
    feat = rng.standard_normal((1, 6, fH, fW, C), dtype=np.float32)
    og = rng.standard_normal((1, 16, 160, 240, C), dtype=np.float32)
    pts = radar_points(rng, 14000)
    frames, t0 = 0, time.perf_counter()
    while True:
        geom = O.get_geometry(fr, rots, trans)
        rb, rd, rf, st, ln = O.voxel_pooling_prepare_v2(geom, dx, bx, nx)
        OC.bev_pool_v2_fwd(depth, feat, rd, rf, rb, (1, 16, 160, 240, C), st, ln, threads=True)
        bp = O.backward_tables(rb, rd, rf)
        OC.bev_pool_v2_bwd(og, depth, feat, bp[1], bp[2], bp[0], bp[3], bp[4], threads=True)
        v, c, n = OC.hard_voxelize(pts, [0.25, 0.25, 8], PC_RANGE, 10, 30000)
        OC.pillar_scatter(np.ascontiguousarray(np.broadcast_to(v.sum(1)[:, :1], (len(v), 64))), np.pad(c, ((0, 0), (1, 0))), 1, 320, 480)
        frames += 1
        el = time.perf_counter() - t0
        if el > budget_s or frames >= 8:
            break
    return {"value": round(frames / el, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{frames} frames of the same bev_ops workload at {res}, B=1 (numpy geometry+rank tables per frame, "
                      f"C/OpenMP pooling fwd+bwd, sequential voxelise, scatter)"}


def cpu_baseline_fusion(res, radar_dims, budget_s=30.0, max_steps=5):
    """The same training step on the host CPU: torch-CPU dense layers + the oracle (port of the
    reference algorithm) for the ops that have no reference CPU kernels.  As in the reference, the
    rank tables are rebuilt every forward and re-sorted every backward (no plan cache)."""
    from omnihd_amd import harness
    from oracle.torch_shim import oracle_ops

    # More threads than ~32 make torch-CPU convolutions at batch 6 slower, not faster (measured on the
    # 256-core GPU host: 435 s/step with 256 threads); the thread count actually used is reported.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)

    with oracle_ops():
        st = harness.FusionTrainStep(res=res, batch=1, radar_dims=radar_dims, device="cpu", dtype="fp32",
                                     channels_last=False, sets=1)
        lss = st.raw_model.lift_splat_shot_vis
        # SURVEY.md 8(d) protocol, bounded: one untimed warm-up step (first-touch allocations, thread pools), then
        # measured steps until the budget is spent (at least 3), median reported
        times, t_all = [], time.perf_counter()
        for k in range(1 + max_steps):
            lss._plans.clear()                     # the reference rebuilds the tables every forward
            t0 = time.perf_counter()
            st.step()
            if k > 0:
                times.append(time.perf_counter() - t0)
            if len(times) >= 3 and time.perf_counter() - t_all > budget_s:
                break
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "step_s": {"median": round(med, 3), "min": round(times[0], 3), "max": round(times[-1], 3)},
            "sample": f"1 warm-up + {len(times)} measured training steps (median) of the same fusion workload at {res}, B=1, fp32: "
                      "torch-CPU dense layers; numpy rank tables rebuilt every forward, C/OpenMP pooling fwd+bwd with per-backward "
                      "re-sort, sequential voxelise, index scatter (reference semantics; the reference has no CPU kernels).  Both legs "
                      "rebuild the tables per step where the calibration changes per step: the GPU leg's per_frame_calibration block "
                      "(the headline value is the static rig of SURVEY 8(d), whose plan is built once)"}


def run_cpu_baseline_child(res, radar_dims, timeout_s=300):
    """The CPU leg runs in a child process (own thread pools, hard wall-clock bound)."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="32", MKL_NUM_THREADS="32", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", res, str(radar_dims)]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        for ln in reversed(out.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"value": None, "unit": "frames/s", "cores": 32, "kind": "port", "sample": "cpu baseline child failed: " + out.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "frames/s", "cores": 32, "kind": "port",
                "sample": f"one CPU training step did not finish within {timeout_s} s (< {1.0 / timeout_s:.4f} frames/s)"}


_T0 = time.time()


def _phase(name):
    """Progress marker on stderr: if the process dies of a GPU fault, the log says in which part."""
    print(f"bench.py phase [{time.time() - _T0:6.1f} s]: {name}", file=sys.stderr, flush=True)


def run_guarded():
    """N = 1: the measurement runs in a CHILD process of this one (the parent never touches the GPU) and its stdout — the one JSON
    line — is passed through.  A child that dies is a failed run: its exit code is this run's exit code (a signal becomes
    128 + signal), nothing is retried.  (Rounds 4-5 restarted a child that died of `Memory access fault by GPU`; the cause was
    found and removed in round 6, profiles/round6/fault_root_cause.txt.)"""
    import subprocess
    env = dict(os.environ, OMNIHD_BENCH_CHILD="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE)
    out = p.stdout.decode(errors="replace")
    sys.stdout.write(out)
    sys.stdout.flush()
    rc = p.returncode
    if rc == 0 and not any(ln.lstrip().startswith("{") for ln in out.splitlines()):
        print("bench.py: the measurement process ended without a result line", file=sys.stderr, flush=True)
        return 1
    if rc != 0:
        print(f"bench.py: the measurement process ended with exit code {rc}", file=sys.stderr, flush=True)
        return 128 - rc if rc < 0 else rc
    return 0


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start N ranks of this script under torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) as CHILD processes — this process has not touched the GPU yet and never does —
    pass rank 0's JSON line through and return the launcher's exit code.  The reference starts its ranks the same way from one
    command (tools/dist_train.sh:7-9: torch.distributed.launch --nproc_per_node=N)."""
    import socket
    import subprocess
    share = os.environ.get("OMNIHD_BENCH_SHARE_GPU") == "1"
    if "--selftest-launch" not in sys.argv and not share and torch.cuda.device_count() < n:      # counting devices does not initialise them
        print(f"bench.py: --gpus {n} but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def selftest_launch(a, world, rank):
    """`--selftest-launch`: the rendezvous / barrier / max-over-ranks / rank-0-prints skeleton of the multi-rank bench with no
    GPU work at all (gloo), so that the launch path of `--gpus N` is testable on a machine without GPUs.  Not a measurement."""
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        comm = {"world_size": dist.get_world_size(), "backend": dist.get_backend()} if world > 1 else None
        print(json.dumps({"selftest": "launch", "n_gpus": world, "max_over_ranks": float(t.item()), "steps": a.steps,
                          "warmup": a.warmup, "comm": comm}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pool_source_hash():
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "omnihd-scenes_amd", "csrc", "bev_pool_v2.hip"), "rb").read()).hexdigest()


def pmc_traffic(res, fwd_kernel):
    """HBM-side bytes per launch of the pooling forward from the committed rocprofv3 PMC passes (profiles/round6/pmc_pool_<res>.json,
    else the round-4 file: FETCH_SIZE x the factor calibrated on a 128 MiB copy + WRITE_SIZE, separate --pmc passes,
    scripts/lab/pmc_bwd.sh) — a pointer to a measurement of THIS kernel, not a counter read in this run: null unless the file names the
    kernel that ran AND was taken from the same csrc/bev_pool_v2.hip (sha256 recorded in the file)."""
    stale = None
    for rnd in ("round6", "round4"):
        rel = f"profiles/{rnd}/pmc_pool_{res}.json"
        pmc = os.path.join(ROOT, rel)
        if not os.path.exists(pmc):
            continue
        rec = json.load(open(pmc))
        k = rec.get("fwd_lean", {})
        if not (fwd_kernel.startswith(rec.get("fwd_kernel", "?")) and "read_bytes_corrected" in k and "write_bytes" in k):
            continue
        if rec.get("pool_source_sha256") != pool_source_hash():
            stale = stale or f"{rel} is stale: csrc/bev_pool_v2.hip changed since the counters were taken"
            continue
        return round(k["read_bytes_corrected"] + k["write_bytes"]), (
            f"{rel} (rocprofv3 --pmc passes of scripts/lab/pmc_bwd.sh on {rec.get('fwd_kernel')}, "
            f"same csrc/bev_pool_v2.hip: sha256 {rec.get('pool_source_sha256', '?')[:12]})")
    return None, stale


def pooling_at_r2(dev, launches):
    """The two pooling kernels at the repo's own resolution (configs[2]: 6 x 544 x 960 images, bevfusion.py:28), isolated
    launches: back-to-back on one buffer set (inputs resident in the Infinity Cache, as behind the depth-head epilogue kernel in
    the step) and after a 512 MiB sweep (cold)."""
    wl = BevOps("r2", 1, dev, seed=1234, sets=2)
    fb, bb = wl.fwd_algorithmic_bytes(), wl.bwd_algorithmic_bytes()
    tf, tb = time_kernel(wl.pool_fwd, 1, launches), time_kernel(wl.pool_bwd, 1, launches)
    cf, cb = time_kernel_cold(wl.pool_fwd, 2, 8), time_kernel_cold(wl.pool_bwd, 2, 8)
    traffic, src = pmc_traffic("r2", wl.fwd_kernel_name())
    out = {"workload": "bev_pool_v2 forward / backward at 544x960 (136x240 feature map, 4 503 872 points, 389 882 intervals)",
           "fwd_kernel": wl.fwd_kernel_name(), "fwd_algorithmic_bytes": fb, "fwd_warm_us": round(tf * 1e6, 2),
           "fwd_frac": round(fb / tf / 1e9 / HBM_PEAK_GBS, 4), "fwd_cold_us": round(cf * 1e6, 2),
           "fwd_frac_cold": round(fb / cf / 1e9 / HBM_PEAK_GBS, 4), "fwd_traffic": traffic, "fwd_traffic_source": src,
           "bwd_kernel": wl.bwd_kernel_name(), "bwd_algorithmic_bytes": bb, "bwd_warm_us": round(tb * 1e6, 2),
           "bwd_frac": round(bb / tb / 1e9 / HBM_PEAK_GBS, 4), "bwd_cold_us": round(cb * 1e6, 2),
           "bwd_frac_cold": round(bb / cb / 1e9 / HBM_PEAK_GBS, 4), "n_points": wl.plan.n_points}
    del wl
    torch.cuda.empty_cache()
    return out


_REAL_STDOUT = None


def emit_line(line):
    """The ONE JSON line, on the process's real stdout (see main: fd 1 is pointed at stderr while the GPU libraries run)."""
    data = (json.dumps(line) + "\n").encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        os.write(1, data)
    else:
        os.write(_REAL_STDOUT, data)


def measure_copy_peak(dev, mib=1024, reps=8):
    """Practical HBM rate of this box: a device-to-device copy of `mib` MiB (read + write = 2x bytes), best of `reps`.  The
    microarchitecture guide quotes 6.29 TB/s for a float4 copy kernel; roofline fractions are reported against the 8 TB/s
    specification AND against this measured figure."""
    n = (mib << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    best = None
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3
        best = t if best is None else min(best, t)
    del a, b
    torch.cuda.empty_cache()
    return round(2.0 * n * 4 / best / 1e9, 1)


def ddp1_child(res, steps, warmup):
    """Fresh process: the fp32 step plain, then inside a one-rank RCCL group under DistributedDataParallel, then plain again —
    each with `warmup` + 7 untimed and `steps` timed steps, per-step events.  The like-for-like pair for `overhead_vs_plain`:
    inside the long bench process the DDP run comes minutes after the headline run (other workloads in between)."""
    import socket
    from omnihd_amd.harness import FusionTrainStep, seed_miopen_db
    seed_miopen_db()
    torch.cuda.set_device(0)
    radar_dims = 7 if res == "r1" else 8
    out = {}

    def run(tag, ddp):
        wl = FusionTrainStep(res=res, batch=1, radar_dims=radar_dims, device="cuda:0", seed=1234, dtype="fp32", ddp=ddp, miopen_find=True)
        for _ in range(warmup + 7):
            wl.step()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for k in range(steps):
            wl.step()
            marks[k + 1].record()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        per = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(steps))
        out[tag] = {"ms_per_step": round(el / steps * 1e3, 4), "median": round(per[len(per) // 2], 3), "p10": round(per[int(0.1 * len(per))], 3),
                    "p90": round(per[min(len(per) - 1, int(0.9 * len(per)))], 3)}
        del wl
        torch.cuda.empty_cache()

    run("plain", False)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        run("ddp", True)
    finally:
        dist.destroy_process_group()
    run("plain_again", False)
    ref = 0.5 * (out["plain"]["median"] + out["plain_again"]["median"])
    out["overhead_median"] = round(out["ddp"]["median"] / ref - 1.0, 4)
    out["overhead_mean"] = round(out["ddp"]["ms_per_step"] / (0.5 * (out["plain"]["ms_per_step"] + out["plain_again"]["ms_per_step"])) - 1.0, 4)
    out["note"] = "fresh child process: plain, one-rank DDP, plain again; overhead = DDP over the mean of the two plain runs"
    return out


def run_ddp1_child(res, steps, warmup, timeout_s=900):
    import subprocess
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--ddp1-child", res, str(steps), str(warmup)],
                           capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, OMNIHD_BENCH_CHILD="1"))
        for ln in reversed(p.stdout.splitlines()):
            if ln.lstrip().startswith("{"):
                return json.loads(ln)
        return {"error": "no line; exit code %d: %s" % (p.returncode, p.stderr[-300:])}
    except subprocess.TimeoutExpired:
        return {"error": f"did not finish within {timeout_s} s"}


def ddp_one_rank(a, dev, local, radar_dims, timed, plain_ms):
    """The SAME fp32 step inside a one-rank RCCL process group under DistributedDataParallel (reducer hooks, bucket views, the
    bucket all-reduce on the communication stream, weight gradients of the side stream written into the bucket views): what
    one GPU can measure of the N > 1 code path.  `overhead_vs_plain` = ms/step over the plain N = 1 step of this run - 1.
    `syncbn_exchange_us`: the naiveSyncBN exchanges of one step (one all-reduce of 2*C floats per layer and direction — a
    one-rank group skips them inside the step, reference ops/norm.py:58) issued back to back in that group."""
    import socket
    from omnihd_amd import ops as ops_mod
    from omnihd_amd.harness import FusionTrainStep, syncbn_exchange_probe
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        wl = FusionTrainStep(res=a.res, batch=a.batch, radar_dims=radar_dims, device=f"cuda:{local}", seed=1234, dtype="fp32",
                             ddp=True, miopen_find=True)
        # set-up: the reducer re-buckets before its second forward, the views settle one pass later — and DDP's logger times
        # every one of a process's first ten iterations with an event SYNCHRONISATION (reducer runtime stats; afterwards every
        # 100th): they are all spent here, so that none of them drains the queue inside the timed steps
        for _ in range(max(4, 11 - a.warmup)):
            wl.step()
        ops_mod.fast_paths_reset()
        el, spread, _ = timed(wl)
        ms = el / a.steps * 1e3
        fp = ops_mod.fast_paths_report()
        probe = syncbn_exchange_probe(wl, iters=20)
        out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ms_per_step": round(ms, 4), "step_ms": spread,
               "plain_ms_per_step": round(plain_ms, 4), "overhead_vs_plain": round(ms / plain_ms - 1.0, 4),
               "wgrad_overlap": fp["wgrad_overlap"], "syncbn_exchange_us": probe,
               "note": "same fp32 training step as the headline inside a 1-rank RCCL group under DistributedDataParallel "
                       "(25 MB buckets, gradient_as_bucket_view); target overhead <= 1 %"}
        del wl
    finally:
        dist.destroy_process_group()
    torch.cuda.empty_cache()
    # The headline step was timed minutes earlier in this process, on a cooler chip (the bf16, library and R2 runs lie between):
    # the plain step once more, right behind the DDP run, is the like-for-like reference (scripts/lab/ddp1_step.py, fresh process
    # per variant: plain 43.8 / DDP 44.1-44.2 ms).  Both overheads are reported.
    try:
        wl = FusionTrainStep(res=a.res, batch=a.batch, radar_dims=radar_dims, device=f"cuda:{local}", seed=1234, dtype="fp32",
                             ddp=False, miopen_find=True)
        for _ in range(3):
            wl.step()
        el2, spread2, _ = timed(wl)
        after = el2 / a.steps * 1e3
        out["plain_after_ms_per_step"] = round(after, 4)
        out["plain_after_step_ms"] = spread2
        out["overhead_vs_plain_after"] = round(ms / after - 1.0, 4)
        out["overhead_vs_plain_after_median"] = round(spread["median"] / spread2["median"] - 1.0, 4)
        del wl
    except Exception as e:      # the block is an extra: never lose the line over it
        out["plain_after_error"] = repr(e)[:200]
    torch.cuda.empty_cache()
    return out


def dense_rooflines(a, world, runs, flops, main_dt):
    """`step_roofline`: dense-layer FLOPs of one step (module graph, harness.count_step_flops) / step time against the dense bf16
    MFMA peak, per precision, with the MFMA work factor stated (the fp32-grade split kernels issue 3 bf16 products per fp32
    product); `conv_roofline`: our implicit-GEMM kernel on the largest convolution from HIP events INSIDE the timed steps;
    `fp32_library`: the fp32 step with every dense convolution on MIOpen's fp32 kernels."""
    out = {}
    cin, cout, k, H, W = CONV_GEOMETRY
    conv_flops = 2.0 * a.batch * H * W * cin * cout * k * k
    sr = {}
    for dt, key in (("fp32", "f32_split_bf16"), ("bf16", "bf16_autocast"), ("fp32_library", "f32_library")):
        if dt not in runs:
            continue
        el = runs[dt][0]
        fl = flops.get("fp32" if dt == "fp32_library" else dt)
        if not fl:
            continue
        tf = fl["total"] * world / (el / a.steps) / 1e12
        factor = 3 if dt == "fp32" else 1
        peak = MFMA_BF16_PEAK_TFLOPS if dt != "fp32_library" else 157.3
        sr[key] = {"flops_per_step": fl["total"], "flops_forward": fl["forward"], "achieved_TFLOPs": round(tf, 1),
                   "peak_TFLOPs": peak * world, "frac": round(tf / (peak * world), 4), "mfma_work_factor": factor,
                   "frac_of_issued_mfma": round(tf * factor / (peak * world), 4)}
    if sr:
        sr["note"] = ("dense layers only (Conv2d / ConvTranspose2d / Linear / DCN contraction: forward + data gradient where the input "
                      "carries one + weight gradient where the weight trains); peak = dense bf16 MFMA 2.5 PFLOP/s (fp32 matrix "
                      "157.3 TFLOP/s for the library line); the split kernels issue 3 bf16 MFMA products per fp32 product "
                      "(mfma_work_factor), so frac_of_issued_mfma is the share of the bf16 pipes' peak they keep busy")
        out["step_roofline"] = sr
    cr = {}
    for dt, kind, key in (("fp32", "conv_split", "f32_split_bf16"), ("bf16", "conv_bf16", "bf16")):
        if dt in runs and runs[dt][2].get(kind):
            t = runs[dt][2][kind]
            factor = 3 if dt == "fp32" else 1
            cr[key] = {"mean_launch_us": round(t * 1e6, 1), "launches": runs[dt][2].get("n_" + kind),
                       "effective_TFLOPs": round(conv_flops / t / 1e12, 1), "mfma_work_factor": factor,
                       "frac_of_bf16_mfma_peak": round(conv_flops * factor / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
    if cr:
        cr["kernel"] = "k_conv_igemm_rs (3x3 implicit GEMM with row-shift reuse of the activation tile, csrc/conv_igemm.hip)"
        cr["geometry"] = f"{cin}->{cout} {k}x{k} @ {H}x{W}, batch {a.batch}: forward and data-gradient launches"
        cr["measured"] = "HIP events on the launching stream around every launch inside the timed steps"
        out["conv_roofline"] = cr
    if "tf32_grade" in runs:
        e4, sp4, _ = runs["tf32_grade"]
        out["tf32_grade"] = {"value": round(a.batch * world * a.steps / e4, 3), "unit": "frames/s", "ms_per_step": round(e4 / a.steps * 1e3, 4),
                             "step_ms": sp4, "precision": "f32 tensors; convolution operands rounded to IEEE half (11 significant bits, as "
                                                          "TF32), f16 x f16 -> f32 MFMA, fp32 accumulation; gradients scaled per tensor by 2^n",
                             "note": "OMNIHD_FP32_CONV=f16: the stride-1 convolutions with 64-multiple channels as ONE IEEE-half MFMA "
                                     "product per fp32 product in all three directions (gradients scaled per tensor by an exact power of "
                                     "two), the precision the reference trains at (TF32 on: tools/train.py:150-153); the other layers "
                                     "stay fp32-grade.  A sibling line, never `value`: the headline keeps three bf16 products per term. "
                                     "Parity: tests/test_conv_f16_gpu.py"}
    if "fp32_library" in runs:
        e3, sp3, _ = runs["fp32_library"]
        out["fp32_library"] = {"value": round(a.batch * world * a.steps / e3, 3), "unit": "frames/s", "ms_per_step": round(e3 / a.steps * 1e3, 4),
                               "step_ms": sp3, "note": "OMNIHD_FP32_CONV=miopen: every dense convolution on MIOpen's fp32 kernels (fp32 MFMA, "
                                                       "no bf16 products); everything else as in the headline step"}
    return out


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--ddp1-child":
        print(json.dumps(ddp1_child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))), flush=True)
        return
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-baseline-child":
        print(json.dumps(cpu_baseline_fusion(sys.argv[2], int(sys.argv[3]))), flush=True)
        return
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus))          # before anything in this process touches the GPU
    profiled = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCP_TOOL", "ROCPROF")) for k in os.environ)
    if (a.gpus == 1 and "WORLD_SIZE" not in os.environ and os.environ.get("OMNIHD_BENCH_CHILD") != "1" and not a.selftest_launch
            and not profiled):                           # (under rocprofv3 the measurement stays in the profiled process)
        raise SystemExit(run_guarded())                  # this process never touches the GPU either
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if a.selftest_launch:
        return selftest_launch(a, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # The driver reads ONE JSON line from stdout.  RCCL prints its version banner on stdout when its first communicator comes up
    # (C code, not sys.stdout), MIOpen and others may do likewise: everything that is not the line goes to stderr from here on,
    # the line itself is written to the real stdout (emit_line).
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    # OMNIHD_BENCH_SHARE_GPU=1: every rank on cuda:0 over gloo — a functional check of the multi-rank path (DDP, SyncBN
    # exchange, max-over-ranks timing) on a one-GPU box; RCCL refuses two ranks on one device.  Not a measurement.
    share = os.environ.get("OMNIHD_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    radar_dims = 7 if a.res == "r1" else 8
    # Dominant north_star kernel (bev_pool_v2 forward): timed FIRST, on the same frame geometry with rotating buffer
    # sets, before the training loop heats the chip (the same kernel inside the step runs ~15 % slower: DVFS
    # after MFMA-heavy convolutions and a polluted L2 — see profiles/ for the in-step rocprofv3 average).
    kernel_times = kernel_cold = other_ops = r2_block = copy_peak = None
    kept_bytes = 0
    if rank == 0:
        _phase("copy peak, pooling kernels isolated")
        copy_peak = measure_copy_peak(dev)
        ops_wl = BevOps(a.res, a.batch, dev, seed=1234)
        if ops_wl.keep_empty:      # same switch (OMNIHD_POOL_KEEP_ZEROS) in the detector's view transformer
            kept_bytes = 4 * ops_wl.C * (ops_wl.plan.n_rows - ops_wl.plan.n_intervals)
        kernel_cold = time_kernel_cold(ops_wl.pool_fwd, len(ops_wl.sets))
        kernel_times = (time_kernel(ops_wl.pool_fwd, len(ops_wl.sets), a.kernel_launches),
                        time_kernel(ops_wl.pool_bwd, len(ops_wl.sets), a.kernel_launches),
                        ops_wl.fwd_algorithmic_bytes(), ops_wl.plan.n_points, ops_wl.plan.n_intervals, ops_wl.fH, ops_wl.fW,
                        ops_wl.fwd_kernel_name(), ops_wl.bwd_kernel_name(), ops_wl.bwd_algorithmic_bytes())
        _phase("voxelisation / pillar scatter isolated")
        other_ops = ops_wl.other_ops(a.res)
        del ops_wl
        torch.cuda.empty_cache()
        if a.res == "r1" and a.batch == 1:
            _phase("pooling kernels at R2")
            r2_block = pooling_at_r2(dev, a.kernel_launches)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(wl):
        """W warm-up steps, then exactly K steps between barrier + synchronize; max over ranks.  An event after every step
        gives the spread of the individual steps (no synchronisation inside the timed region)."""
        # Python's cyclic collector: this process has built (and dropped) several detectors by the time the later blocks run, and a
        # full collection then walks that whole long-lived heap — 119 / 189 ms inside ONE timed step of the one-rank DDP block on two
        # leases (profiles/round6/bench_r6_lease*.json, always the same step: the trigger counts allocations).  The heap of everything
        # built so far is collected once and frozen (gc.freeze: out of the collector's sight) before the timed steps; the collector
        # stays ON, what it costs on the objects the steps themselves create is inside the time, and is reported ("gc").  Done in
        # front of the WARM-UP steps: the collection takes ~0.1 s in which the GPU runs dry, and the warm-up refills its queue (done
        # behind them, the first timed step of every block ran 5-7 ms long: profiles/round6/bench_r6_lease3.json).
        import gc
        gc.collect()
        gc.freeze()
        for _ in range(a.warmup):
            wl.step()
        gc_log, gc_t0 = [], [0.0]

        def gc_watch(phase, info):
            if phase == "start":
                gc_t0[0] = time.perf_counter()
            else:
                gc_log.append((info["generation"], (time.perf_counter() - gc_t0[0]) * 1e3))
        gc.callbacks.append(gc_watch)
        barrier()
        import omnihd_amd.plan as plan_mod
        from omnihd_amd import ops as ops_mod
        plan_mod.TIMING = []                 # events around every pooling kernel launched inside the timed steps
        ops_mod.conv_kernels.CONV_TIMING, ops_mod.conv_kernels.CONV_TIMING_GEOMETRY = [], CONV_GEOMETRY   # ... and around our kernel on the largest convolution
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for k in range(a.steps):
            wl.step()
            marks[k + 1].record()
        barrier()
        el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        gc.callbacks.remove(gc_watch)
        gc.unfreeze()
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        raw = [marks[k].elapsed_time(marks[k + 1]) for k in range(a.steps)]
        per = sorted(raw)
        q = lambda f: per[min(len(per) - 1, int(f * len(per)))]
        slow = [(k, round(v, 2)) for k, v in enumerate(raw) if v > 1.08 * q(0.5)]     # which of the timed steps ran long (index, ms)
        pool = {}
        for kind, e0, e1 in plan_mod.TIMING:
            pool.setdefault(kind, []).append(e0.elapsed_time(e1) * 1e-3)
        plan_mod.TIMING = None
        for kind, e0, e1 in ops_mod.conv_kernels.CONV_TIMING:
            pool.setdefault("conv_" + kind, []).append(e0.elapsed_time(e1) * 1e-3)
        ops_mod.conv_kernels.CONV_TIMING = None
        in_step = {k: sum(v) / len(v) for k, v in pool.items() if v}
        in_step.update({"n_" + k: len(v) for k, v in pool.items()})
        return float(el.item()), {"median": round(q(0.5), 3), "p10": round(q(0.1), 3), "p90": round(q(0.9), 3), "max": round(per[-1], 3),
                                  "slow_steps": slow,
                                  "gc": {"collections": len(gc_log), "full": sum(1 for g, _ in gc_log if g == 2),
                                         "ms_total": round(sum(t for _, t in gc_log), 2), "ms_max": round(max([t for _, t in gc_log] or [0.0]), 2)}}, in_step

    runs, flops, comm, fast, ddp1, r2_step, per_frame, tf32_error = {}, {}, None, {}, None, None, None, None
    if a.workload == "fusion":
        from omnihd_amd.harness import FusionTrainStep
        from omnihd_amd import ops as ops_mod_
        for dt in (["fp32", "bf16"] if a.dtype == "both" else [a.dtype]):
            _phase(f"training step, {dt}: build + set-up steps")
            wl = FusionTrainStep(res=a.res, batch=a.batch, radar_dims=radar_dims, device=f"cuda:{local}", seed=1234 + rank,
                                 dtype=dt, ddp=world > 1, miopen_find=True)
            # set-up, not warm-up: MIOpen's find step and the per-geometry weight-gradient measurement run during the first
            # two or three steps (each geometry once); they are finished before the W warm-up steps start
            for _ in range(3):
                wl.step()
            wl.sync_choices()                 # N > 1: every rank runs the kernels rank 0 measured best
            if world > 1:
                # the reducer's bucket views settle one pass after it re-buckets; and the first ten iterations of a DDP process
                # each carry an event synchronisation of its logger (see ddp_one_rank): none of them inside the timed steps
                for _ in range(max(1, 8 - a.warmup)):
                    wl.step()
            ops_mod_.fast_paths_reset()
            _phase(f"training step, {dt}: warm-up + timed steps")
            runs[dt] = timed(wl)
            fast[dt] = ops_mod_.fast_paths_report()      # live counters of the timed steps (+ warm-up), not the switches
            if dt == ("fp32" if a.dtype == "both" else a.dtype) and os.environ.get("OMNIHD_BENCH_PER_FRAME", "1") != "0":
                # the same step with a NEW camera calibration every step, as on the reference's own frames (lidar2img is composed
                # per sample from the ego poses: datasets/newscenes_dataset.py:203-216): every step is a plan-cache miss
                _phase(f"training step, {dt}: a new calibration every step")
                from omnihd_amd import pool_plan as pp_mod
                built0 = pp_mod.BUILDS["device_plans"]
                wl.jitter_calibration = True
                per_frame = timed(wl) + (pp_mod.BUILDS["device_plans"] - built0,)
                wl.jitter_calibration = False
            _phase(f"training step, {dt}: counting the step's FLOPs (one eval-mode forward)")
            # every rank counts (one eval-mode forward: no collective, no BatchNorm statistics touched): the ranks' control flow
            # stays identical, whatever a module of a future config does in its forward
            from omnihd_amd.harness import count_step_flops
            flops[dt] = count_step_flops(wl)
            if world > 1 and dt == ("fp32" if a.dtype == "both" else a.dtype):
                from omnihd_amd.harness import comm_report
                comm = comm_report(wl)        # after the timed region: its no_sync steps let the ranks' weights drift
            del wl
            torch.cuda.empty_cache()
        if a.dtype == "both" and world == 1:
            # the same fp32 step with every dense convolution on MIOpen's fp32 kernels (strict fp32 MFMA, no bf16 products):
            # the "library" line beside the split-bf16 headline
            os.environ["OMNIHD_FP32_CONV"] = "miopen"
            _phase("training step, fp32 on the library's kernels")
            try:
                wl = FusionTrainStep(res=a.res, batch=a.batch, radar_dims=radar_dims, device=f"cuda:{local}", seed=1234 + rank,
                                     dtype="fp32", ddp=False, miopen_find=True)
                for _ in range(3):
                    wl.step()
                runs["fp32_library"] = timed(wl)
                del wl
            finally:
                os.environ.pop("OMNIHD_FP32_CONV", None)
            torch.cuda.empty_cache()
            if os.environ.get("OMNIHD_BENCH_TF32_GRADE", "1") != "0":
                # the sibling line the reference's own precision asks for: it trains with TF32 left on (tools/train.py:150-153), i.e.
                # 11 significant bits per operand — one IEEE-half MFMA product per fp32 product, fp32 accumulation (never `value`)
                os.environ["OMNIHD_FP32_CONV"] = "f16"
                _phase("training step, fp32 tensors with the convolutions in the TF32-grade half form")
                try:
                    wl = FusionTrainStep(res=a.res, batch=a.batch, radar_dims=radar_dims, device=f"cuda:{local}", seed=1234 + rank,
                                         dtype="fp32", ddp=False, miopen_find=True)
                    for _ in range(max(3, 8 - a.warmup)):
                        wl.step()
                    runs["tf32_grade"] = timed(wl)
                    del wl
                except Exception as e:      # the block is an extra: never lose the line over it
                    runs.pop("tf32_grade", None)
                    tf32_error = repr(e)[:200]
                finally:
                    os.environ.pop("OMNIHD_FP32_CONV", None)
                torch.cuda.empty_cache()
        main_dt = "fp32" if a.dtype == "both" else a.dtype
        if a.dtype == "both" and world == 1 and a.res == "r1" and a.batch == 1:
            # BASELINE configs[2]: the repo's own resolution (544x960, 8 radar channels, bevfusion.py:28,164) — 5 fp32 steps
            _phase("training step at R2, fp32")
            wl = FusionTrainStep(res="r2", batch=1, radar_dims=8, device=f"cuda:{local}", seed=1234, dtype="fp32", ddp=False, miopen_find=True)
            for _ in range(3):
                wl.step()
            keep = (a.steps, a.warmup)
            a.steps, a.warmup = 5, 2
            try:
                e_r2, sp_r2, in_r2 = timed(wl)
                r2_step = {"ms_per_step": round(e_r2 / a.steps * 1e3, 4), "value": round(a.steps / e_r2, 3), "unit": "frames/s", "steps": a.steps,
                           "step_ms": sp_r2, "fwd_in_step_us": round(in_r2.get("fwd", 0.0) * 1e6, 2), "bwd_in_step_us": round(in_r2.get("bwd", 0.0) * 1e6, 2)}
            finally:
                a.steps, a.warmup = keep
            del wl
            torch.cuda.empty_cache()
        if a.dtype in ("both", "fp32") and world == 1 and os.environ.get("OMNIHD_BENCH_DDP1", "1") != "0":
            _phase("training step in a one-rank process group")
            ddp1 = ddp_one_rank(a, dev, local, radar_dims, timed, runs["fp32"][0] / a.steps * 1e3)
            if a.res == "r1" and a.batch == 1 and os.environ.get("OMNIHD_BENCH_DDP1_FRESH", "1") != "0":
                _phase("the same pair (plain / one-rank DDP / plain) in a fresh process")
                ddp1["fresh_process"] = run_ddp1_child(a.res, a.steps, a.warmup)
    else:
        runs["f32"] = timed(BevOps(a.res, a.batch, dev, seed=1234 + rank))
        main_dt = "f32"
    el, spread, in_step = runs[main_dt]

    if rank == 0:
        t_fwd, t_bwd, fwd_bytes, n_points, n_intervals, fH, fW, fwd_kernel, bwd_kernel, bwd_bytes = kernel_times
        cold_fwd = kernel_cold
        # `achieved` is computed from the launches INSIDE the timed steps when the workload has them (events around the kernel
        # on its launching stream); the isolated loops are reported beside it
        t_step = in_step.get("fwd") if a.workload == "fusion" else None
        t_main = t_step if t_step else t_fwd
        ach = fwd_bytes / t_main / 1e9
        # HBM-side bytes per launch from rocprofv3 PMC passes (scripts/pmc_traffic.sh): FETCH_SIZE x the factor
        # calibrated on a 128 MiB read of the same width (2.0 on gfx950, as the microarch guide says) + WRITE_SIZE
        # (a pointer to a committed measurement of THIS kernel, not a counter read in this run: PMC passes need rocprofv3)
        traffic, traffic_src = pmc_traffic(a.res, fwd_kernel) if a.batch == 1 else (None, None)
        # BEV rows no frustum point reaches keep the zeros of the previous launch (plan.py::_kept_output): the kernel skips
        # their zero fill, so it MOVES fewer bytes than the SURVEY 8(d) formula counts — both rates are reported
        kept = kept_bytes
        ach_moved = (fwd_bytes - kept) / t_main / 1e9
        line = {
            "metric": "frames/sec (6-cam+6-radar BEV fwd+bwd)", "value": round(a.batch * world * a.steps / el, 3),
            "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "step_ms": spread, "vs_baseline": None, "dtype": ("bf16" if main_dt == "bf16" else "f32"),
            "data": "synthetic",
            "config": {"workload": (f"fusion@{a.res}: BEVFUSION_depth (reference config bevfusion_NewScenes/bevfusion.py) training step "
                                    f"fwd+bwd+clip+AdamW+refresh of the kernels' weight images; 6 cams {RES[a.res][0]}x{RES[a.res][1]}, R50+FPNC, LSS D=59 C=64, BEV 240x160x16, "
                                    f"radar N~U(8k,20k)x{radar_dims}, 30 GT boxes; random-init weights"
                                    if a.workload == "fusion" else
                                    f"bev_ops@{a.res}: LSS bev_pool_v2 fwd(dense)+bwd, radar hard-voxelize + pillar scatter; "
                                    f"6 cams {RES[a.res][0]}x{RES[a.res][1]} -> fmap {fH}x{fW}, D=59, C=64, BEV 240x160x16; "
                                    "conv backbone/BEV encoder NOT in this workload"),
                       "precision": ("fp32 activations and master weights; dense convolutions on the fp32-grade split-bf16 MFMA kernels of this "
                                     "library (x*w = hi*hi + hi*lo + lo*hi with fp32 accumulation, 5e-6 of an fp32 convolution: parity 1e-4 in "
                                     "tests/test_conv_split_gpu.py) or on MIOpen's fp32 kernels, measured per geometry (OMNIHD_FP32_CONV)"
                                     if (main_dt != "bf16" and a.workload == "fusion") else
                                     "bf16 autocast for the dense convolutions, everything else fp32" if a.workload == "fusion" else "fp32 operators"),
                       "frames_per_gpu": a.batch, "n_points": n_points, "n_intervals": n_intervals,
                       "streams": ("radar branch in line on the step's stream (OMNIHD_DUAL_STREAM=%s; the second stream of rounds 3-5 was the "
                                   "trigger of the intermittent GPU memory fault, profiles/round6/fault_root_cause.txt); weight gradients of the "
                                   "split convolutions behind the pooling backward on a side stream, joined at the end of backward "
                                   "(OMNIHD_WGRAD_OVERLAP=%s, fp32 only: %s) — kernels measured inside the step run 3-5 %% above their "
                                   "isolated durations" % (
                                       os.environ.get("OMNIHD_DUAL_STREAM", "0"), os.environ.get("OMNIHD_WGRAD_OVERLAP", "1"),
                                       "active" if (main_dt != "bf16" and os.environ.get("OMNIHD_WGRAD_OVERLAP", "1") != "0"
                                                    and os.environ.get("OMNIHD_FP32_CONV", "tune") != "miopen") else "inactive")
                                   if a.workload == "fusion" else "one stream"),
                       "parallelism": (f"dp{world}: one rank per GPU, DDP gradient all-reduce over RCCL (25 MB buckets, "
                                       "overlapped with backward) + naiveSyncBN stat exchange" if a.workload == "fusion"
                                       else f"dp{world} (independent frames, no data-path collective)")},
            "roofline": {"kernel": fwd_kernel + " (bev_pool_v2 forward, dense, balanced tiles, azimuth XCD schedule, rows without points keep their zeros)", "bound": "hbm",
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": fwd_bytes, "mean_launch_us": round(t_main * 1e6, 2),
                         "zero_rows_not_rewritten_bytes": kept, "achieved_on_moved_bytes": round(ach_moved, 1),
                         "frac_on_moved_bytes": round(ach_moved / HBM_PEAK_GBS, 4),
                         "measured": ("launches inside the %d timed steps (HIP events on the launching stream)" % a.steps
                                      if t_step else "isolated back-to-back launches"),
                         "isolated_cache_warm_us": round(t_fwd * 1e6, 2), "isolated_cold_us": round(cold_fwd * 1e6, 2),
                         "bwd_kernel": bwd_kernel, "bwd_mean_launch_us": round((in_step.get("bwd") or t_bwd) * 1e6, 2),
                         "bwd_isolated_us": round(t_bwd * 1e6, 2),
                         "bwd_frac": round(bwd_bytes / (in_step.get("bwd") or t_bwd) / 1e9 / HBM_PEAK_GBS, 4),
                         "bwd_algorithmic_bytes": bwd_bytes,
                         # the same rates against what this box's memory system delivers to a plain copy (measured above) and
                         # against the microarchitecture guide's float4-copy figure
                         "copy_peak_measured": copy_peak, "copy_peak_guide": 6290.0,
                         "frac_vs_copy_peak": round(ach / copy_peak, 4), "frac_vs_guide_copy_peak": round(ach / 6290.0, 4),
                         "frac_on_moved_bytes_vs_copy_peak": round(ach_moved / copy_peak, 4),
                         "bwd_frac_of_copy_peak": round(bwd_bytes / (in_step.get("bwd") or t_bwd) / 1e9 / copy_peak, 4),
                         "bwd_in_step_over_isolated": round((in_step.get("bwd") or t_bwd) / t_bwd, 3)},
        }
        line["ops_roofline"] = other_ops
        if r2_block is not None:
            if r2_step is not None:
                r2_block["step"] = r2_step
                r2_block["step_ms"] = r2_step["ms_per_step"]
            r2_block["fwd_frac_vs_copy_peak"] = round(r2_block["fwd_algorithmic_bytes"] / (r2_block["fwd_warm_us"] * 1e-6) / 1e9 / copy_peak, 4)
            r2_block["bwd_frac_vs_copy_peak"] = round(r2_block["bwd_algorithmic_bytes"] / (r2_block["bwd_warm_us"] * 1e-6) / 1e9 / copy_peak, 4)
            line["r2"] = r2_block
        if per_frame is not None:
            e_pf, sp_pf, in_pf, built = per_frame
            line["per_frame_calibration"] = {
                "value": round(a.batch * world * a.steps / e_pf, 3), "unit": "frames/s", "ms_per_step": round(e_pf / a.steps * 1e3, 4),
                "step_ms": sp_pf, "over_cached_step": round(e_pf / el, 4),
                "plans_built": built, "plan_builds_in_timed_steps": in_pf.get("n_plan", 0),
                "plan_build_us": round(in_pf.get("plan", 0.0) * 1e6, 1),
                "fwd_in_step_us": round(in_pf.get("fwd", 0.0) * 1e6, 2), "bwd_in_step_us": round(in_pf.get("bwd", 0.0) * 1e6, 2),
                "note": "the headline step with lidar2img perturbed by ego-motion-sized jitter (<= 1 deg yaw, 0.5 m) in EVERY step, so "
                        "the plan cache misses by construction — what the reference's own frames do to it "
                        "(cam_stream_lss_bevpoolv2_depthnet.py:283-300 rebuilds the rank tables every forward).  A miss = one "
                        "library call that enqueues the plan's ~25 launches (csrc/pool_plan.hip): no host read-back, no "
                        "synchronisation; plan_build_us = device time between HIP events around those launches inside the step; "
                        "the kept output buffer is handed from calibration to calibration and only rows that emptied are "
                        "zero-filled.  The cpu_baseline leg rebuilds its tables every step too"}
        if fast:
            line["fast_paths"] = fast.get(main_dt) if len(fast) == 1 else fast
        if ddp1 is not None:
            line["ddp_1rank"] = ddp1
        from omnihd_amd import ops as ops_mod
        line["kernel_choice_table"] = ops_mod.choice_table_info()
        if a.workload == "fusion":
            line.update(dense_rooflines(a, world, runs, flops, main_dt))
            if tf32_error:
                line["tf32_grade"] = {"error": tf32_error}
        if comm is not None:
            line["comm"] = comm
        if a.workload == "fusion" and a.dtype == "both":
            e2, sp2, _ = runs["bf16"]
            line["bf16_autocast"] = {"value": round(a.batch * world * a.steps / e2, 3), "unit": "frames/s",
                                     "ms_per_step": round(e2 / a.steps * 1e3, 4), "step_ms": sp2,
                                     "note": "dense convolutions under bf16 autocast (our implicit-GEMM MFMA kernels for forward / data / weight gradient or MIOpen, measured per geometry), "
                                             "pooling / voxelisation / losses fp32; deviation from the fp32 step bounded in "
                                             "tests/test_detector_gpu.py::test_bf16_step_deviation_from_the_fp32_step"}
        if not a.no_cpu_baseline and world == 1:      # rank 0 at N=1 only: at N>1 the other ranks would sit in the
            if a.workload == "fusion":                # closing barrier for minutes while the host cores are busy
                line["cpu_baseline"] = run_cpu_baseline_child(a.res, radar_dims)
            else:
                line["cpu_baseline"] = cpu_baseline(a.res)
        emit_line(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
