#!/usr/bin/env python3
"""Tile-size sweep of the one-table pooling forward (azimuth XCD schedule), in one run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
from omnihd_amd import ops, plan as P

res = sys.argv[1] if len(sys.argv) > 1 else "r1"
wl = bench.BevOps(res, 1, torch.device("cuda:0"), 1234)
nbytes = wl.fwd_algorithmic_bytes()
D, fhw = wl.D, wl.fH * wl.fW
descs = {}


def run(s):
    depth, feat, og, out, dg, fg, tb = wl.sets[s]
    ops.bev_pool_v2_forward_lean(depth, feat, tb[0], tb[2], cur[s], out, D, fhw)


for items, long_len in ((192, 192), (256, 256), (320, 256), (384, 256), (384, 384), (448, 384), (512, 384), (512, 512), (640, 512), (768, 512), (896, 384)):
    tiles = ops.csr_tiles(wl.plan.row_ptr, items, long_len)
    order = P.tile_schedule(wl.plan.row_ptr, tiles, wl.plan.ranks_feat, (wl.fH, wl.fW), grid=wl.plan.grid, layout="byxz")
    desc = ops.tile_descriptors(wl.plan.row_ptr, tiles, order)
    cur = [desc.clone() for _ in wl.sets]
    ts = [bench.time_kernel(run, len(wl.sets), 60) for _ in range(2)]
    print(f"{res} W={items:5d} L={long_len:4d}: {min(ts)*1e6:6.1f} us  frac {nbytes/min(ts)/8e12:.3f}  tiles {tiles.numel()-1}")
