import sys, numpy as np
res = sys.argv[1]
t = np.load('/tmp/pool_locality_tables_%s.npz' % res)
row, rd, rf, ptr = t['row'].astype(np.int64), t['rd'].astype(np.int64), t['rf'].astype(np.int64), t['ptr'].astype(np.int64)
X, Y, Z = 240, 160, 16
ncell = X * Y
cell_of_pt = row // Z
cpts = np.bincount(cell_of_pt, minlength=ncell)
cy, cx = np.divmod(np.arange(ncell), X)
ang = np.arctan2(cy - 79.5, cx - 119.5); rng = np.hypot(cy - 79.5, cx - 119.5) * 0.5
def evaluate(cell_order, T, name):
    # cut ordered cells into tiles of ~T points (+16 rows per cell)
    w = cpts[cell_order] + 16
    cw = np.cumsum(w); tile_of_cell_sorted = cw // T
    tile_of_cell = np.empty(ncell, np.int64); tile_of_cell[cell_order] = tile_of_cell_sorted
    tp = tile_of_cell[cell_of_pt]
    nt = tile_of_cell_sorted.max() + 1
    # distinct (tile, pixel) pairs
    pair = tp * (1 << 22) + rf
    up = np.unique(pair)
    pix_per_tile = np.bincount(up >> 22, minlength=nt)
    pts_per_tile = np.bincount(tp, minlength=nt)
    m = pts_per_tile > 0
    print('%-34s tiles %5d  pts/tile %6.0f  pix/tile mean %6.1f p90 %6.0f max %5d  reuse(total) %.2f  staged MB %.1f' % (
        name, nt, pts_per_tile[m].mean(), pix_per_tile[m].mean(), np.percentile(pix_per_tile[m], 90), pix_per_tile.max(),
        len(rf) / len(up), len(up) * 256 / 1e6))
# current: cells in (y,x) order
for T in (768, 1536, 3072):
    evaluate(np.arange(ncell), T, 'yx-runs T=%d' % T)
for nsec in (360, 720, 1440, 2880):
    sec = np.floor((ang + np.pi) / (2 * np.pi) * nsec).astype(np.int64)
    order = np.lexsort((rng, sec))
    for T in (768, 1536, 3072):
        evaluate(order, T, 'polar %d sectors T=%d' % (nsec, T))
