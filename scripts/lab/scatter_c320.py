#!/usr/bin/env python3
"""Channels-last pillar scatter with C = 320 in a fresh process (round 5: the parametrised test failed at C = 320 only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import numpy as np, torch
from omnihd_amd import ops
from oracle import cpu as OC
dev = torch.device("cuda:0")
for C in (320, 256, 192, 512):
    rng = np.random.default_rng(C)
    B, ny, nx = 1, 96, 160
    for rep in range(2):
        cells = rng.permutation(ny * nx)[:6000]
        coors = np.stack([np.zeros(6000, int), np.zeros(6000, int), cells // nx, cells % nx], 1).astype(np.int32)
        feats = rng.standard_normal((6000, C), dtype=np.float32)
        got = ops.pillar_scatter(torch.from_numpy(feats).to(dev), torch.from_numpy(coors).to(dev), B, ny, nx, channels_last=True).cpu().numpy()
        want = OC.pillar_scatter(feats, coors, B, ny, nx)
        bad = np.argwhere(got != want)
        print("C", C, "rep", rep, "mismatches", len(bad), "of", got.size, "nan", int(np.isnan(got).sum()))
        if len(bad):
            ch = np.unique(bad[:, 1]); print("   channels", ch[:20], "... n", len(ch), " first", bad[:3].tolist(),
                                             "got", got[tuple(bad[0])], "want", want[tuple(bad[0])])
