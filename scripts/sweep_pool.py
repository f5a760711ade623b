#!/usr/bin/env python3
"""GPU micro-benchmark of the pooling kernels (not part of the bench contract): tile-size sweep,
tiled vs untiled dense forward, backward; rotating buffer sets, HIP-event timing."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from omnihd_amd import ops, plan as P  # noqa: E402


def main():
    res = sys.argv[1] if len(sys.argv) > 1 else "r1"
    dev = torch.device("cuda:0")
    wl = bench.BevOps(res, 1, dev, 1234)
    nbytes = wl.fwd_algorithmic_bytes()
    print(f"{res}: points {wl.plan.n_points} intervals {wl.plan.n_intervals} rows {wl.plan.n_rows} alg bytes {nbytes/1e6:.1f} MB")
    wl.tiled = False
    t = bench.time_kernel(wl.pool_fwd, len(wl.sets), 40)
    print(f"untiled dense fwd : {t*1e6:8.1f} us  {nbytes/t/1e9:7.0f} GB/s")
    wl.tiled = True
    for items, long_len in ((256, 256), (384, 384), (512, 512), (768, 512), (1024, 256), (896, 384), (640, 640)):
        tiles = ops.csr_tiles(wl.plan.row_ptr, items, long_len)
        order = P.tile_schedule(wl.plan.row_ptr, tiles, wl.plan.ranks_feat, (wl.fH, wl.fW))
        order_az = P.tile_schedule(wl.plan.row_ptr, tiles, wl.plan.ranks_feat, (wl.fH, wl.fW), grid=wl.plan.grid, layout="byxz")
        res_t = []
        for o in (None, order, order_az):
            desc = ops.tile_descriptors(wl.plan.row_ptr, tiles, o)
            for s in wl.sets:
                s[6][8] = desc.clone()
            res_t.append(bench.time_kernel(wl.pool_fwd, len(wl.sets), 40))
        print(f"tiled fwd W={items:5d} L={long_len:4d}: banded {res_t[0]*1e6:7.1f} us {nbytes/res_t[0]/1e9:6.0f} GB/s | "
              f"column {res_t[1]*1e6:7.1f} us {nbytes/res_t[1]/1e9:6.0f} GB/s | azimuth {res_t[2]*1e6:7.1f} us {nbytes/res_t[2]/1e9:6.0f} GB/s  tiles {tiles.numel()-1}")
    wl.sched_bwd = False
    t = bench.time_kernel(wl.pool_bwd, len(wl.sets), 40)
    print(f"bwd reference-API kernel (incl. 2 memsets): {t*1e6:8.1f} us")
    wl.sched_bwd = True
    t = bench.time_kernel(wl.pool_bwd, len(wl.sets), 40)
    print(f"bwd scheduled kernel (incl. depth_grad memset): {t*1e6:8.1f} us")
    lin = P.pixel_schedule(wl.plan.bp_ranks_feat, wl.plan.bp_starts, wl.plan.bp_lengths, wl.N * wl.fH * wl.fW, None)
    for s in wl.sets:
        s[6][10] = lin.clone()
    t = bench.time_kernel(wl.pool_bwd, len(wl.sets), 40)
    print(f"bwd scheduled kernel, linear pixel order      : {t*1e6:8.1f} us")
    # plain device copy of the same byte count as a local ceiling
    a = torch.empty(nbytes // 8, dtype=torch.float32, device=dev)
    bufs = [(torch.empty_like(a), torch.empty_like(a)) for _ in range(4)]
    t = bench.time_kernel(lambda k: bufs[k][1].copy_(bufs[k][0]), 4, 40)
    print(f"torch copy of {nbytes/2e6:.0f} MB -> {nbytes/2e6:.0f} MB: {t*1e6:8.1f} us  {nbytes/t/1e9:7.0f} GB/s")


if __name__ == "__main__":
    main()
