"""Pooling BACKWARD: how often do the pixels of a 16-pixel patch ask for the same out_grad row?  points / distinct (patch, voxel row)
for 16x1 runs, 8x2 and 4x4 blocks (the case for k_pool_bwd_stream).  Tables: build_tables.py."""
import sys, numpy as np
res = sys.argv[1]
t = np.load('/tmp/pool_locality_tables_%s.npz' % res)
row, rd, rf = t['row'].astype(np.int64), t['rd'].astype(np.int64), t['rf'].astype(np.int64)
H, W = {'r1': (64, 176), 'r2': (136, 240)}[res]
fhw = H * W; D = 59
img = rf // fhw; hw = rf % fhw; h, w = hw // W, hw % W
d = (rd // fhw) % D
for name, patch in (('16x1 run', (img * H + h) * ((W + 15) // 16) + w // 16), ('4x4 block', (img * ((H + 3) // 4) + h // 4) * ((W + 3) // 4) + w // 4),
                    ('32x1 run', (img * H + h) * ((W + 31) // 32) + w // 32), ('8x2 block', (img * ((H + 1) // 2) + h // 2) * ((W + 7) // 8) + w // 8)):
    key = patch * (1 << 24) + row
    u = np.unique(key)
    # per (patch, depth): distinct rows
    key2 = (patch * 64 + d) * (1 << 24) + row
    u2 = np.unique(key2)
    # wave-level (4 adjacent pixels of a run, same depth)
    print('%s %-10s points %d | distinct (patch,voxel) %d -> reuse %.2f | distinct (patch,depth,voxel) %d -> reuse %.2f' % (res, name, len(row), len(u), len(row) / len(u), len(u2), len(row) / len(u2)))
wave = ((img * H + h) * ((W + 3) // 4) + w // 4)
u3 = np.unique((wave * 64 + d) * (1 << 24) + row)
print('%s 4x1 (one wave) same depth: reuse %.2f' % (res, len(row) / len(u3)))
