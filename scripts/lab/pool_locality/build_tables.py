import sys, numpy as np, torch, time
sys.path[:0] = ['/root/repo', '/root/repo/omnihd-scenes_amd']
from oracle import lss_oracle as O
res = sys.argv[1] if len(sys.argv) > 1 else 'r2'
H, W, fx = {'r1': (256, 704, 410.0), 'r2': (544, 960, 560.0)}[res]
dx, bx, nx = O.gen_dx_bx([-60, 60, .5], [-40, 40, .5], [-3, 5, .5])
fr = O.create_frustum((H, W), 4, [1., 60., 1.])
l2i = O.synthetic_rig(H, W, fx)
inv = [torch.Tensor(m).inverse() for m in l2i]
rots = torch.stack([m[:3, :3] for m in inv])[None].numpy(); trans = torch.stack([m[:3, 3] for m in inv])[None].numpy()
geom = O.get_geometry(fr, rots, trans)
t0 = time.time()
rb, rd, rf, st, ln = O.voxel_pooling_prepare_v2(geom, dx, bx, nx)
print('tables', time.time() - t0, 'npts', len(rb), 'nint', len(st))
X, Y, Z = map(int, nx)
D, fH, fW = fr.shape[:3]
# byxz row order
z = rb // (X * Y); y = (rb // X) % Y; x = rb % X
row = (y * X + x) * Z + z
o = np.argsort(row, kind='stable'); row, rd, rf = row[o], rd[o], rf[o]
nrows = X * Y * Z
cnt = np.bincount(row, minlength=nrows)
ptr = np.concatenate([[0], np.cumsum(cnt)])
np.savez('/tmp/pool_locality_tables_%s.npz' % res, row=row.astype(np.int32), rd=rd, rf=rf, ptr=ptr.astype(np.int32))
