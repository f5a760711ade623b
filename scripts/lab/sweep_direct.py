#!/usr/bin/env python3
"""Tile-size sweep of the direct pooling forward (k_pool_fwd_direct has no LDS record window, so tiles may be any size)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
from omnihd_amd import ops, plan as P

res = sys.argv[1] if len(sys.argv) > 1 else "r1"
wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
nbytes = wl.fwd_algorithmic_bytes()
D, fhw = wl.D, wl.fH * wl.fW
pl = wl.plan
for s in range(len(wl.sets)):
    wl.sets[s][3].zero_()
for items, long_len in ((512, 512), (768, 512), (1024, 512), (1024, 1024), (1280, 1024), (1536, 1024), (2048, 1024), (2048, 2048), (3072, 2048), (4096, 4096)):
    tiles = ops.csr_tiles(pl.row_ptr, items, long_len)
    order = P.tile_schedule(pl.row_ptr, tiles, pl.ranks_feat, (wl.fH, wl.fW), grid=pl.grid, layout="byxz")
    desc = ops.tile_descriptors(pl.row_ptr, tiles, order)
    dt = P.direct_tables_from(pl.ranks_row, pl.ranks_depth, tiles, desc)
    dts = [[t.clone() for t in dt] for _ in wl.sets]

    def run(s):
        depth, feat, og, out, dg, fg, tb = wl.sets[s]
        ops.bev_pool_v2_forward_direct(depth, feat, dts[s][0], dts[s][1], dts[s][2], tb[2], out, D, fhw, empty_rows_kept=True)

    t1 = min(bench.time_kernel(run, 1, 60) for _ in range(2))
    t4 = min(bench.time_kernel(run, 4, 60) for _ in range(2))
    print(f"{res} items={items:5d} long={long_len:4d} tiles {tiles.numel()-1:5d}: warm {t1*1e6:6.1f} us ({nbytes/t1/8e12:.3f})  4 sets {t4*1e6:6.1f} us ({nbytes/t4/8e12:.3f})")
