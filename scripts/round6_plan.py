"""Round 6: what a NEW calibration costs — the device-side plan build (csrc/pool_plan.hip) against the host-scheduled
omnihd_amd.plan.build_plan, and the pooling kernels on a device-built plan (capacity grid / exact grid / host plan).
Usage: python scripts/round6_plan.py [r1|r2|both] [--iters N]   (prints one JSON line per resolution)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "omnihd-scenes_amd")):
    sys.path.insert(0, p)

import omnihd_amd  # noqa: E402
from omnihd_amd import ops, plan as P, pool_plan  # noqa: E402
from omnihd_amd.harness import synthetic_lidar2img  # noqa: E402
from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet import LiftSplatShoot_Depth  # noqa: E402

PC = [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]


def events_us(fn, n, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, host


def jitter(l2i, rng):
    a = np.radians(rng.uniform(-1.0, 1.0))
    T = np.eye(4)
    T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    T[:3, 3] = rng.uniform(-0.5, 0.5, 3) * [1, 1, 0.1]
    inv = torch.Tensor(np.stack([m @ T for m in l2i])).inverse()
    return inv[:, :3, :3][None].contiguous(), inv[:, :3, 3][None].contiguous()


def run(res, iters):
    H, W = {"r1": (256, 704), "r2": (544, 960)}[res]
    dev = torch.device("cuda:0")
    net = LiftSplatShoot_Depth(final_dim=(H, W), camera_depth_range=[1, 60, 1], pc_range=PC, downsample=4, grid=0.5, inputC=256,
                               camC=64, norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01))
    net.frustum.data = net.frustum.data.to(dev)
    l2i = np.asarray(synthetic_lidar2img(res), dtype=np.float64)
    rng = np.random.default_rng(0)
    rigs = [tuple(x.to(dev) for x in jitter(l2i, rng)) for _ in range(8)]
    axes = net._frustum_axes(dev)
    dx, bx, nx = net.dx.numpy(), net.bx.numpy(), net.nx.numpy()
    k = [0]

    def build_dev():
        r, tr = rigs[k[0] % len(rigs)]
        k[0] += 1
        return pool_plan.build_device_plan(dx, bx, nx, rots=r, trans=tr, axes=axes)

    def build_host():
        r, tr = rigs[k[0] % len(rigs)]
        k[0] += 1
        geom = net.get_geometry(r, tr).contiguous().float()
        origin = tr[..., :2].float().mean(dim=(0, 1)).cpu().tolist()
        return omnihd_amd.build_plan(geom, dx, bx, nx, layout="byxz", origin_xy=origin)

    out = {"res": res}
    out["device_build_us"], out["device_build_host_us"] = events_us(build_dev, iters)
    t0 = time.perf_counter()
    for _ in range(3):
        hp = build_host()
        torch.cuda.synchronize()
    out["host_build_wall_us"] = (time.perf_counter() - t0) / 3 * 1e6
    r, tr = rigs[0]
    dp = pool_plan.build_device_plan(dx, bx, nx, rots=r, trans=tr, axes=axes)
    geom = net.get_geometry(r, tr).contiguous().float()
    hp = omnihd_amd.build_plan(geom, dx, bx, nx, layout="byxz", origin_xy=tr[..., :2].float().mean(dim=(0, 1)).cpu().tolist())
    out["counts"] = dp.counts(wait=True)
    out["capacity"] = dict(tiles_cap=dp.tiles_cap, patch_per=dp.patch_per, n_patch=dp.n_patch, workspace_mb=dp.workspace_bytes / 2 ** 20)
    depth = torch.rand(1, 6, 59, H // 4, W // 4, device=dev)
    feat = torch.randn(1, 6, H // 4, W // 4, 64, device=dev)
    o = torch.empty(dp.n_rows, 64, device=dev)
    og = torch.randn(dp.n_rows, 64, device=dev)
    dg, fg = torch.empty_like(depth), torch.empty_like(feat)

    def fwd_dev(slots):
        saved = dp.launch_slots
        dp.launch_slots = lambda: slots
        try:
            return events_us(lambda: pool_plan._forward_direct_dev(depth, feat, dp, o, 0, None), 50)[0]
        finally:
            dp.launch_slots = saved

    pt, ivl_rel, desc32 = P.direct_tables(hp)
    out["fwd_us"] = dict(
        device_plan_capacity_grid=fwd_dev(dp.tiles_cap),
        device_plan_exact_grid=fwd_dev(8 * out["counts"]["tiles_per_xcd"]),
        host_plan=events_us(lambda: ops.bev_pool_v2_forward_direct(depth, feat, pt, ivl_rel, desc32, hp.row_ptr, o, 59,
                                                                   (H // 4) * (W // 4)), 50)[0])
    out["bwd_us"] = dict(
        device_plan=events_us(lambda: ops.bev_pool_v2_backward_patch(og, depth, feat, None, dp.row_bin, dp.pix_ptr, dp.patch_order,
                                                                     dg, fg), 50)[0],
        host_plan=events_us(lambda: ops.bev_pool_v2_backward_patch(og, depth, feat, None, P._row_bin(hp), hp.pix_ptr,
                                                                   hp.patch_order, dg, fg), 50)[0])
    po = dp.patch_order.cpu().numpy().reshape(8, -1)
    hpo = hp.patch_order.cpu().numpy().reshape(8, -1)
    out["patch_runs"] = dict(device=[int((x >= 0).sum()) for x in po], host=[int((x >= 0).sum()) for x in hpo])
    print(json.dumps(out))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "both"
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
    for res in (("r1", "r2") if which == "both" else (which,)):
        run(res, iters)
