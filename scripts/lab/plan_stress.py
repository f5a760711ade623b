"""Stress of the device-built pooling plan: many calibrations in a row, each used at once (forward with the kept output buffer handed
on from the previous calibration, backward), compared bit for bit with the host-built plan of the same geometry.
Usage: python scripts/lab/plan_stress.py [iters] [H W fx]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "omnihd-scenes_amd")):
    sys.path.insert(0, p)
import omnihd_amd  # noqa: E402
from omnihd_amd.plan import planned_pool  # noqa: E402
from oracle import lss_oracle as O  # noqa: E402

PC = [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    H, W, fx = (int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (256, 704, 410.0)
    dev = torch.device("cuda:0")
    dx, bx, nx = O.gen_dx_bx([PC[0], PC[3], 0.5], [PC[1], PC[4], 0.5], [PC[2], PC[5], 0.5])
    fr = torch.from_numpy(O.create_frustum((H, W), 4, [1, 60, 1])).to(dev)
    axes = tuple(torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dev) for a in O.frustum_axes((H, W), 4, [1, 60, 1]))
    l2i = O.synthetic_rig(H, W, fx)
    rng = np.random.default_rng(1)
    fH, fW = H // 4, W // 4
    bad = 0
    for it in range(iters):
        a = np.radians(rng.uniform(-1.5, 1.5))
        T = np.eye(4)
        T[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        T[:3, 3] = rng.uniform(-1.0, 1.0, 3) * [1, 1, 0.1]
        inv = torch.Tensor(np.stack([m @ T for m in l2i])).inverse()
        rots, trans = inv[:, :3, :3][None].contiguous().to(dev), inv[:, :3, 3][None].contiguous().to(dev)
        depth = torch.rand(1, 6, 59, fH, fW, device=dev)
        feat = torch.randn(1, 6, fH, fW, 64, device=dev)
        dp = omnihd_amd.build_device_plan(dx, bx, nx, rots=rots, trans=trans, axes=axes)
        res = []
        for plan, keep in ((dp, True), (None, False)):
            if plan is None:
                p0 = fr[..., 0] * fr[..., 2]
                p1 = fr[..., 1] * fr[..., 2]
                p2 = fr[..., 2]
                R = rots.view(1, 6, 1, 1, 1, 3, 3)
                geom = torch.stack([(R[..., k, 0] * p0 + R[..., k, 1] * p1) + R[..., k, 2] * p2 for k in range(3)], -1) + trans.view(1, 6, 1, 1, 1, 3)
                plan = omnihd_amd.build_plan(geom.contiguous(), dx, bx, nx, layout="byxz")
            d, f = depth.clone().requires_grad_(), feat.clone().requires_grad_()
            out = planned_pool(d, f, plan, keep_empty_rows=keep)
            w = torch.linspace(0.5, 1.5, out.numel(), device=dev).view(out.shape[0], -1)
            (out.reshape(out.shape[0], -1) * w).sum().backward()
            res.append((out.detach().clone(), d.grad.clone(), f.grad.clone()))
            del out
        ok = [bool(torch.equal(x, y)) for x, y in zip(*res)]
        if not all(ok):
            bad += 1
            print(f"iteration {it}: out / depth_grad / feat_grad equal: {ok}; max diffs",
                  [float((x - y).abs().max()) for x, y in zip(*res)], dp.counts(wait=True), flush=True)
    print(f"PLAN_STRESS {H}x{W}: {bad} mismatching iterations of {iters}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
