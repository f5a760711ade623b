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
def evaluate(cell_order, T, name, wins=(256, 128)):
    w = cpts[cell_order] + 16
    cw = np.cumsum(w); tile_sorted = cw // T
    tile_of_cell = np.empty(ncell, np.int64); tile_of_cell[cell_order] = tile_sorted
    tp = tile_of_cell[cell_of_pt]
    nt = tile_sorted.max() + 1
    o = np.argsort(tp, kind='stable'); tps, rfs, rds = tp[o], rf[o], rd[o]
    tptr = np.searchsorted(tps, np.arange(nt + 1))
    per = nt // 8
    out = []
    for win in wins:
        fs, ds = [], []
        for k in (0, 1, 3):
            for s in range(k * per, (k + 1) * per - win + 1, 96):
                a, b = tptr[s], tptr[s + win]
                fs.append(len(np.unique(rfs[a:b])) * 256 / 1e6); ds.append(len(np.unique(rds[a:b] // 32)) * 128 / 1e6)
        out.append('win%d feat %.2f/%.2f depth %.2f' % (win, np.mean(fs), np.max(fs), np.mean(ds)))
    print('%-30s tiles %5d | %s' % (name, nt, ' | '.join(out)))
for nsec in (64, 128, 256, 512, 1024):
    sec = np.floor((ang + np.pi) / (2 * np.pi) * nsec).astype(np.int64)
    evaluate(np.lexsort((rng, sec)), 768, 'polar %d sectors' % nsec)
    # serpentine: alternate range direction per sector
    r2 = np.where(sec % 2 == 0, rng, -rng)
    evaluate(np.lexsort((r2, sec)), 768, 'polar %d serpentine' % nsec)
print('--- camera-centric azimuth')
yaws = np.radians([0, 60, -60, 180, 120, -120])
px, py = (cx - 119.5) * 0.5, (cy - 79.5) * 0.5       # metric cell centres (approx)
zone = np.argmin(np.abs(((ang[:, None] - yaws[None, :]) + np.pi) % (2 * np.pi) - np.pi), axis=1)
camx, camy = np.cos(yaws)[zone], np.sin(yaws)[zone]
ang_c = np.arctan2(py - camy, px - camx)
rel = ((ang_c - yaws[zone]) + np.pi) % (2 * np.pi) - np.pi        # azimuth around the zone's camera relative to its axis
# global key: zone order around the circle, then relative azimuth
zone_rank = np.argsort(np.argsort(yaws % (2 * np.pi)))[zone]
rngc = np.hypot(py - camy, px - camx)
for nsec in (16, 32, 64):
    sec = zone_rank * nsec + np.clip(np.floor((rel + np.pi / 6 + 0.2) / (np.pi / 3 + 0.4) * nsec), 0, nsec - 1).astype(np.int64)
    evaluate(np.lexsort((rngc, sec)), 768, 'camera-centric %d/zone' % nsec)
key = zone_rank * 10.0 + rel
evaluate(np.argsort(key, kind='stable'), 768, 'camera-centric pure azimuth')
