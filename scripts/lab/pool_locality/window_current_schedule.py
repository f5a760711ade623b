import sys, numpy as np
res = sys.argv[1]
t = np.load('/tmp/pool_locality_tables_%s.npz' % res)
row, rd, rf, ptr = t['row'], t['rd'].astype(np.int64), t['rf'].astype(np.int64), t['ptr'].astype(np.int64)
X, Y, Z = 240, 160, 16
nrows = X * Y * Z
H, W = {'r1': (64, 176), 'r2': (136, 240)}[res]
T, L = 768, 512
cnt = ptr[1:] - ptr[:-1]
r = np.arange(nrows)
start = np.zeros(nrows, bool); start[0] = True
item = r + ptr[:-1]
start[1:] |= (item[1:] // T) != (item[:-1] // T)
lng = cnt > L
start |= lng; start[1:] |= lng[:-1]
tile_row = np.concatenate([np.nonzero(start)[0], [nrows]])
nt = len(tile_row) - 1
print('tiles', nt)
mid = np.minimum((tile_row[:-1] + tile_row[1:]) // 2, nrows - 1)
cell = mid // Z; yy, xx = (cell // X) % Y, cell % X
key = np.arctan2(yy - (Y - 1) / 2., xx - (X - 1) / 2.)
order = np.argsort(key, kind='stable')
lo, hi = ptr[tile_row[:-1]], ptr[tile_row[1:]]
work = (hi - lo) + (tile_row[1:] - tile_row[:-1])
cw = np.cumsum(work[order]); run = np.minimum((cw * 8 // (cw[-1] + 1)), 7)
def window_stats(tiles, win=256, step=64, name=''):
    fs, ds, dl64 = [], [], []
    for s in range(0, max(1, len(tiles) - win + 1), step):
        tl = tiles[s:s + win]
        idx = np.concatenate([np.arange(lo[t], hi[t]) for t in tl])
        fs.append(len(np.unique(rf[idx])) * 256 / 1e6)
        ds.append(len(np.unique(rd[idx] // 32)) * 128 / 1e6)
        dl64.append(len(idx))
    print(name, 'window feature MB: mean %.2f max %.2f | depth-line MB mean %.2f max %.2f | pts/window %.0f' % (np.mean(fs), np.max(fs), np.mean(ds), np.max(ds), np.mean(dl64)))
for k in range(8):
    tiles = order[run == k]
    idx = np.concatenate([np.arange(lo[t], hi[t]) for t in tiles])
    print('xcd', k, 'tiles', len(tiles), 'pts', len(idx), 'feat MB %.2f' % (len(np.unique(rf[idx])) * 256 / 1e6), 'depth lines MB %.2f' % (len(np.unique(rd[idx] // 32)) * 128 / 1e6))
    if k in (0, 3):
        for win in (256, 128, 64):
            window_stats(tiles, win, 64, ' win%d' % win)
