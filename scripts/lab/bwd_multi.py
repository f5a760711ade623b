#!/usr/bin/env python3
"""LAB: patch backward with several CONSECUTIVE patches per workgroup (cost-balanced contiguous runs, spatial order) vs the
product kernel.  OMNIHD_POOL_BWD_MULTI=W (workgroups per XCD) is read once per process by the lab build of the library.
Usage: bwd_multi.py [r1|r2] tag [fixed_cost]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
from omnihd_amd import plan as P
res, tag = sys.argv[1], sys.argv[2]
fixed = int(sys.argv[3]) if len(sys.argv) > 3 else 150
W = int(os.environ.get("OMNIHD_POOL_BWD_MULTI", "0"))
wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
plan = wl.plan
if W > 0:
    n_img, fH, fW = wl.batch * wl.N, wl.fH, wl.fW
    fhw = fH * fW
    ppi = (fhw + 15) // 16
    p = torch.arange(n_img * ppi)
    img, k = p // ppi, p % ppi
    h, w = (k * 16) // fW, (k * 16) % fW
    key = ((img * ((fH + 3) // 4) + h // 4) * ((fW + 15) // 16 + 1) + w // 16) * 4 + h % 4
    order = p[torch.argsort(key, stable=True)]
    pp = plan.pix_ptr.cpu().long()
    lens = (pp[1:] - pp[:-1]).view(n_img, fhw)
    cost = (lens.view(n_img, ppi, 16).sum(-1).view(-1) + fixed)[order].double()
    cum = torch.cumsum(cost, 0)
    cuts = [0] + torch.searchsorted(cum, cum[-1] * torch.arange(1, 8, dtype=torch.float64) / 8).tolist() + [order.numel()]
    runs = [order[cuts[i]:cuts[i + 1]] for i in range(8)]
    per = W + 1 + max(r.numel() for r in runs)
    tab = torch.full((8, per), -1, dtype=torch.int32)
    for i, r in enumerate(runs):
        c = cost[cuts[i]:cuts[i + 1]]
        cc = torch.cumsum(c, 0)
        inner = torch.searchsorted(cc, cc[-1] * torch.arange(1, W, dtype=torch.float64) / W)
        ptr = torch.cat([torch.zeros(1, dtype=torch.long), inner, torch.tensor([r.numel()])])
        tab[i, :W + 1] = ptr.int()
        tab[i, W + 1:W + 1 + r.numel()] = r.int()
    plan.patch_order = tab.view(-1).contiguous().to(plan.pix_ptr.device)
    wl.sets = [s[:6] + (tuple(list(s[6][:12]) + [plan.patch_order.clone()]),) for s in wl.sets]
    sizes = (ptr[1:] - ptr[:-1])
    print(tag, "last XCD run: patches per workgroup min/mean/max", int(sizes.min()), float(sizes.float().mean()), int(sizes.max()))
depth, feat, og, out, dg, fg, tb = wl.sets[0]
dg.fill_(float("nan")); fg.fill_(float("nan"))
wl.pool_bwd(0)
torch.cuda.synchronize()
ref = f"/tmp/bwd_multi_{res}.pt"
if os.path.exists(ref):
    a_dg, a_fg = torch.load(ref)
    print(tag, res, "depth_grad identical", bool(torch.equal(a_dg, dg.cpu())), "feat_grad identical", bool(torch.equal(a_fg, fg.cpu())), flush=True)
else:
    torch.save((dg.cpu(), fg.cpu()), ref)
    print(tag, res, "stored reference gradients; finite", bool(torch.isfinite(dg).all() and torch.isfinite(fg).all()), flush=True)
nb = wl.bwd_algorithmic_bytes()
ts = [bench.time_kernel(wl.pool_bwd, len(wl.sets), 60) for _ in range(4)]
print(tag, res, "MULTI=%s fixed=%d" % (os.environ.get("OMNIHD_POOL_BWD_MULTI", "0"), fixed), " ".join(f"{t*1e6:6.1f} us ({nb/t/8e12:.3f})" for t in ts), flush=True)
