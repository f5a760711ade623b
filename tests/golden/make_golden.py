#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own Python.

Runs only in the authoring container (it needs /root/reference); nothing from the reference is
copied into the repo — the outputs are input/expected-output arrays (.npz).  The reference
package cannot be imported as a whole (mmcv / mmdet / mmdet3d / torchvision are absent and
``rcfusion/voxel_encoders/pillar_encoder.py:8`` imports a name that does not exist), so the
needed files are imported one by one behind inert stand-in *modules* for the absent third-party
packages (the stand-ins provide decorators/builders only; every line of arithmetic executed is
the reference's).

The one native dependency, ``bev_pool_v2_ext``, has no CPU implementation in the reference; the
fixtures that go through ``QuickCumsumCuda`` therefore route the two ext calls to oracle/
(liboracle.so).  Those fixtures pin the reference's PYTHON logic (sorting, interval building,
gradient routing); the kernel arithmetic itself is pinned by the reference's known-answer test
(values restated in tests/test_oracle.py) and by oracle/_ref on the GPU.

Usage:  python tests/golden/make_golden.py
"""
import ctypes
import importlib
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch
from torch import nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")


# ---------------------------------------------------------------------------------------------
# inert stand-ins for the absent third-party packages
# ---------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _passthrough_decorator(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]
    return lambda fn: fn


def _build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    t = cfg.pop("type")
    cfg.pop("requires_grad", None)
    cls = {"BN": nn.BatchNorm2d, "BN1d": nn.BatchNorm1d, "BN2d": nn.BatchNorm2d,
           "naiveSyncBN1d": nn.BatchNorm1d, "naiveSyncBN2d": nn.BatchNorm2d}[t]
    return "bn" + str(postfix), cls(num_features, **cfg)


class _Registry:
    def register_module(self, *a, **k):
        return _passthrough_decorator(*a, **k)


def install_stubs():
    _mod("mmcv")
    _mod("mmcv.runner", force_fp32=_passthrough_decorator, auto_fp16=_passthrough_decorator)
    _mod("mmcv.cnn", build_norm_layer=_build_norm_layer, build_conv_layer=None, ConvModule=None,
         NORM_LAYERS=_Registry())
    _mod("mmdet")
    _mod("mmdet.models")
    _mod("mmdet.models.backbones")
    _mod("mmdet.models.backbones.resnet", BasicBlock=None)
    _mod("torchvision")
    _mod("torchvision.models")
    _mod("torchvision.models.resnet", resnet18=None)
    _mod("torchvision.utils", save_image=None)
    _mod("mmdet3d")
    _mod("mmdet3d.models")
    _mod("mmdet3d.models.fusion_layers", apply_3d_transformation=None)
    _mod("mmdet3d.models.builder", VOXEL_ENCODERS=_Registry())
    _mod("mmdet3d.ops", DynamicScatter=None)
    # the reference package tree, as namespace-like shells over the real directories
    for name, rel in [("projects", "projects"),
                      ("projects.mmdet3d_plugin", "projects/mmdet3d_plugin"),
                      ("projects.mmdet3d_plugin.ops", "projects/mmdet3d_plugin/ops"),
                      ("projects.mmdet3d_plugin.ops.bev_pool_v2", "projects/mmdet3d_plugin/ops/bev_pool_v2"),
                      ("projects.mmdet3d_plugin.utils", "projects/mmdet3d_plugin/utils"),
                      ("projects.mmdet3d_plugin.bevfusion", "projects/mmdet3d_plugin/bevfusion"),
                      ("projects.mmdet3d_plugin.bevfusion.detectors", "projects/mmdet3d_plugin/bevfusion/detectors"),
                      ("projects.mmdet3d_plugin.rcfusion", "projects/mmdet3d_plugin/rcfusion"),
                      ("projects.mmdet3d_plugin.rcfusion.voxel_encoders", "projects/mmdet3d_plugin/rcfusion/voxel_encoders"),
                      ]:
        m = _mod(name)
        m.__path__ = [os.path.join(REF, rel)]


def load_ref(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


# ---------------------------------------------------------------------------------------------
# bev_pool_v2_ext stand-in -> oracle C kernels (the reference has no CPU kernels)
# ---------------------------------------------------------------------------------------------
def install_ext_via_oracle():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))

    def p(t):
        return ctypes.c_void_p(t.data_ptr())

    def bev_pool_v2_forward(depth, feat, out, ranks_depth, ranks_feat, ranks_bev,
                            interval_lengths, interval_starts):
        lib.oracle_bev_pool_v2_fwd(ctypes.c_int(feat.size(4)), ctypes.c_int(interval_lengths.size(0)),
                                   p(depth), p(feat), p(ranks_depth), p(ranks_feat), p(ranks_bev),
                                   p(interval_starts), p(interval_lengths), p(out))

    def bev_pool_v2_backward(out_grad, depth_grad, feat_grad, depth, feat, ranks_depth, ranks_feat,
                             ranks_bev, interval_lengths, interval_starts):
        lib.oracle_bev_pool_v2_bwd(ctypes.c_int(out_grad.size(4)), ctypes.c_int(interval_lengths.size(0)),
                                   p(out_grad), p(depth), p(feat), p(ranks_depth), p(ranks_feat),
                                   p(ranks_bev), p(interval_starts), p(interval_lengths),
                                   p(depth_grad), p(feat_grad))

    _mod("projects.mmdet3d_plugin.ops.bev_pool_v2.bev_pool_v2_ext",
         bev_pool_v2_forward=bev_pool_v2_forward, bev_pool_v2_backward=bev_pool_v2_backward)


def lss_shell(cls, final_dim, downsample, dbound, pc_range, grid):
    """An attribute bag carrying exactly what the reference methods read from ``self``
    (LiftSplatShoot_Depth.__init__ :165-191), without building the conv networks."""
    gen_dx_bx = sys.modules[cls.__module__].gen_dx_bx
    s = types.SimpleNamespace()
    s.grid_conf = {"xbound": [pc_range[0], pc_range[3], grid],
                   "ybound": [pc_range[1], pc_range[4], grid],
                   "zbound": [pc_range[2], pc_range[5], grid],
                   "dbound": dbound}
    s.final_dim = final_dim
    dx, bx, nx = gen_dx_bx(s.grid_conf["xbound"], s.grid_conf["ybound"], s.grid_conf["zbound"])
    s.dx, s.bx, s.nx = dx, bx, nx
    s.downsample = downsample
    s.fH, s.fW = final_dim[0] // downsample, final_dim[1] // downsample
    s.frustum = cls.create_frustum(s).data
    s.D = s.frustum.shape[0]
    return s


def rig_rots_trans(lidar2img):
    """Exactly bevf_faster_rcnn_bevdepth.py:121-124: torch.Tensor(mat).inverse() per camera."""
    rots, trans = [], []
    for mat in lidar2img:
        mat = torch.Tensor(mat)
        rots.append(mat.inverse()[:3, :3])
        trans.append(mat.inverse()[:3, 3].view(-1))
    return torch.stack(rots)[None], torch.stack(trans)[None]


def main():
    install_stubs()
    install_ext_via_oracle()
    lss = load_ref("projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet",
                   "projects/mmdet3d_plugin/bevfusion/detectors/cam_stream_lss_bevpoolv2_depthnet.py")
    bp = sys.modules["projects.mmdet3d_plugin.ops.bev_pool_v2.bev_pool"]
    cls = lss.LiftSplatShoot_Depth
    from oracle import lss_oracle as O

    out = {}
    # ---- G1: grid constants + frustum axes at the two resolutions and a tiny one ------------
    pc_range = [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]
    for tag, fd in [("r1", (256, 704)), ("r2", (544, 960)), ("tiny", (32, 48))]:
        s = lss_shell(cls, fd, 4, [1, 60, 1], pc_range, 0.5)
        fr = s.frustum.numpy()
        xs, ys, ds = fr[0, 0, :, 0], fr[0, :, 0, 1], fr[:, 0, 0, 2]
        assert np.array_equal(fr[..., 0], np.broadcast_to(xs[None, None, :], fr.shape[:3]))
        assert np.array_equal(fr[..., 1], np.broadcast_to(ys[None, :, None], fr.shape[:3]))
        assert np.array_equal(fr[..., 2], np.broadcast_to(ds[:, None, None], fr.shape[:3]))
        out[f"g1_{tag}_xs"], out[f"g1_{tag}_ys"], out[f"g1_{tag}_ds"] = xs, ys, ds
        out[f"g1_{tag}_dx"], out[f"g1_{tag}_bx"], out[f"g1_{tag}_nx"] = s.dx.numpy(), s.bx.numpy(), s.nx.numpy()

    # ---- G2/G3/G4: tiny rig: geometry, prepare tables, backward tables ----------------------
    torch.manual_seed(7)
    tiny_range = [-8.0, -6.0, -1.0, 8.0, 6.0, 1.0]
    s = lss_shell(cls, (32, 48), 4, [1.0, 9.0, 1.0], tiny_range, 1.0)
    l2i = O.synthetic_rig(32, 48, 30.0, yaws_deg=(0, 90, 180), radius=0.5, height=0.3)
    rots1, trans1 = rig_rots_trans(l2i)
    rots = torch.cat([rots1, rots1.flip(1)], 0)      # B=2, second sample uses another camera order
    trans = torch.cat([trans1, trans1.flip(1)], 0)
    geom = cls.get_geometry(s, rots, trans)
    tabs = cls.voxel_pooling_prepare_v2(s, geom)
    out["g2_rots"], out["g2_trans"] = rots.numpy(), trans.numpy()
    out["g2_geom"] = geom.numpy()
    out["g2_final_dim"] = np.array([32, 48]); out["g2_dbound"] = np.array([1.0, 9.0, 1.0])
    out["g2_pc_range"] = np.array(tiny_range); out["g2_grid"] = np.array(1.0)
    names = ["ranks_bev", "ranks_depth", "ranks_feat", "starts", "lengths"]
    for k, t in zip(names, tabs):
        out[f"g3raw_{k}"] = t.numpy()            # exactly what the reference returned
    # torch's argsort is not stable for small inputs (reference defect D6): the order INSIDE an
    # interval is implementation-defined.  The canonical form sorts every interval by ranks_depth.
    canon = np.lexsort((tabs[1].numpy(), tabs[0].numpy()))
    raw_is_canonical = bool(np.array_equal(canon, np.arange(canon.size)))
    for k, t in zip(names[:3], tabs[:3]):
        out[f"g3_{k}"] = t.numpy()[canon]
    out["g3_starts"], out["g3_lengths"] = tabs[3].numpy(), tabs[4].numpy()
    out["g3_raw_is_canonical"] = np.array(raw_is_canonical)
    print("tiny rig:", geom.shape, "kept", tabs[0].numel(), "intervals", tabs[3].numel(),
          "raw order canonical:", raw_is_canonical)

    # adversarial coordinates on a 4x4x2 grid (SURVEY Appendix C, D3 check) laid out as (1,1,1,1,W,3)
    s2 = lss_shell(cls, (32, 48), 4, [1.0, 9.0, 1.0], [-2.0, -2.0, -1.0, 2.0, 2.0, 1.0], 1.0)
    adv = torch.tensor([[-2.5, -1.5, -0.5], [-3.0, -1.5, -0.5], [1.999, 1.999, 0.999],
                        [2.0, 0.0, 0.0], [-2.0, -2.0, -1.0], [0.0, 0.0, 0.0],
                        [-0.0, -0.0, -0.0], [1.0, 1.0, 0.5], [1.0, 1.0, 0.5],
                        [float("nan"), 0.0, 0.0], [1e30, 0.0, 0.0], [-1e30, 0.0, 0.0],
                        [0.5, -2.9999, 0.0], [0.5, -3.0001, 0.0], [1.0, 1.0, 0.5]],
                       dtype=torch.float32).view(1, 1, 1, 1, -1, 3)
    tabs_adv = cls.voxel_pooling_prepare_v2(s2, adv)
    out["g3adv_coor"] = adv.numpy()
    canon = np.lexsort((tabs_adv[1].numpy(), tabs_adv[0].numpy()))
    for k, t in zip(names[:3], tabs_adv[:3]):
        out[f"g3adv_{k}"] = t.numpy()[canon]
    out["g3adv_starts"], out["g3adv_lengths"] = tabs_adv[3].numpy(), tabs_adv[4].numpy()
    print("adversarial:", [t.tolist() for t in tabs_adv])

    # ---- G4 + pooling through the reference's autograd Function (ext -> oracle) -------------
    B, N, D, H, W = geom.shape[:5]
    C = 8
    depth = torch.rand(B, N, D, H, W).softmax(2).requires_grad_()
    feat = torch.randn(B, N, H, W, C).requires_grad_()
    nx = s.nx
    shape = (B, int(nx[2]), int(nx[1]), int(nx[0]), C)
    bev = bp.bev_pool_v2(depth, feat, tabs[1], tabs[2], tabs[0], shape, tabs[3], tabs[4])
    w = torch.randn_like(bev)
    (bev * w).sum().backward()
    out["g4_depth"], out["g4_feat"] = depth.detach().numpy(), feat.detach().numpy()
    out["g4_bev"], out["g4_w"] = bev.detach().numpy(), w.numpy()
    out["g4_depth_grad"], out["g4_feat_grad"] = depth.grad.numpy(), feat.grad.numpy()
    # backward tables exactly as QuickCumsumCuda.backward builds them (bev_pool.py:47-57)
    order = tabs[2].argsort()
    rf, rd, rb = tabs[2][order], tabs[1][order], tabs[0][order]
    kept = torch.ones(rb.shape[0], dtype=torch.bool)
    kept[1:] = rf[1:] != rf[:-1]
    st = torch.where(kept)[0].int()
    ln = torch.zeros_like(st)
    ln[:-1] = st[1:] - st[:-1]
    ln[-1] = rb.shape[0] - st[-1]
    out["g4raw_bp_ranks_bev"], out["g4raw_bp_ranks_depth"], out["g4raw_bp_ranks_feat"] = rb.numpy(), rd.numpy(), rf.numpy()
    canon = np.lexsort((rd.numpy(), rb.numpy(), rf.numpy()))     # canonical: by (feat, bev, depth)
    out["g4_bp_ranks_bev"], out["g4_bp_ranks_depth"], out["g4_bp_ranks_feat"] = rb.numpy()[canon], rd.numpy()[canon], rf.numpy()[canon]
    out["g4_bp_starts"], out["g4_bp_lengths"] = st.numpy(), ln.numpy()

    # ---- full-size checksums on the synthetic rig (R1 and R2, B=1) --------------------------
    for tag, (Hh, Ww, fx) in [("r1", (256, 704, 410.0)), ("r2", (544, 960, 560.0))]:
        s = lss_shell(cls, (Hh, Ww), 4, [1, 60, 1], pc_range, 0.5)
        l2i = O.synthetic_rig(Hh, Ww, fx)
        rots, trans = rig_rots_trans(l2i)
        geom = cls.get_geometry(s, rots, trans)
        g_or = O.get_geometry(s.frustum.numpy(), rots.numpy(), trans.numpy())
        exact = np.array_equal(g_or, geom.numpy())
        maxdiff = float(np.abs(g_or - geom.numpy()).max())
        tabs = cls.voxel_pooling_prepare_v2(s, geom)
        tabs_or = O.voxel_pooling_prepare_v2(g_or, s.dx.numpy(), s.bx.numpy(), s.nx.numpy())
        same_tables = all(np.array_equal(a.numpy(), b) for a, b in zip(tabs, tabs_or))
        print(f"{tag}: oracle geometry bit-exact vs reference torch-CPU: {exact} (max |diff| {maxdiff:.3e});"
              f" tables from oracle geometry identical: {same_tables}")
        out[f"full_{tag}_rots"], out[f"full_{tag}_trans"] = rots.numpy(), trans.numpy()
        out[f"full_{tag}_lidar2img"] = l2i
        cs = np.array([tabs[0].numel(), tabs[3].numel()] +
                      [int(t.long().sum()) for t in tabs] +
                      [int(tabs[4].max())], dtype=np.int64)
        out[f"full_{tag}_checksums"] = cs   # n_pts, n_int, sum(rb), sum(rd), sum(rf), sum(starts), sum(lengths), max_len
        out[f"full_{tag}_geom_sum"] = geom.double().sum(dim=(0, 1, 2, 3, 4)).numpy()
        out[f"full_{tag}_oracle_geom_exact"] = np.array(exact)
        out[f"full_{tag}_oracle_tables_same"] = np.array(same_tables)
        # sorted stability check: ranks_depth ascending inside every interval
        rd = tabs[1].numpy(); stt = tabs[3].numpy()
        head = np.zeros(rd.shape[0], bool); head[stt] = True
        assert np.all((np.diff(rd) > 0) | head[1:]), "reference argsort was not stable on this data"
        print(tag, "checksums", cs.tolist())

    # ---- G5: gaussian depth target -----------------------------------------------------------
    gz = load_ref("projects.mmdet3d_plugin.utils.gaussian", "projects/mmdet3d_plugin/utils/gaussian.py")
    torch.manual_seed(11)
    dm = torch.zeros(1, 2, 32, 48)
    mask = torch.rand(1, 2, 32, 48) < 0.08
    dm[mask] = torch.rand(int(mask.sum())) * 12.0
    tgt, mind = gz.generate_guassian_depth_target(dm, 4, [1.0, 9.0, 1.0], constant_std=0.5)
    out["g5_depth_map"], out["g5_target"], out["g5_min_depth"] = dm.numpy(), tgt.numpy(), mind.numpy()

    # ---- G6: pillar feature nets with fixed weights (BN in eval) -----------------------------
    vu = load_ref("projects.mmdet3d_plugin.rcfusion.voxel_encoders.utils",
                  "projects/mmdet3d_plugin/rcfusion/voxel_encoders/utils.py")
    vu.PFNLayer_Radar_vod = None   # defect D1: name imported by pillar_encoder.py:8 does not exist
    pe = load_ref("projects.mmdet3d_plugin.rcfusion.voxel_encoders.pillar_encoder",
                  "projects/mmdet3d_plugin/rcfusion/voxel_encoders/pillar_encoder.py")
    torch.manual_seed(3)
    vsz, pcr = [0.25, 0.25, 8], [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]
    net = pe.PillarFeatureNetV1(in_channels=8, feat_channels=[64], with_distance=False, voxel_size=vsz,
                                point_cloud_range=pcr, norm_cfg=dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01))
    bn = net.pfn_layers[0].norm
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
    net.eval()
    M, P, Fd = 37, 10, 8
    npts = torch.randint(1, P + 1, (M,), dtype=torch.int32)
    coors = torch.stack([torch.zeros(M, dtype=torch.int32), torch.zeros(M, dtype=torch.int32),
                         torch.randint(0, 320, (M,), dtype=torch.int32),
                         torch.randint(0, 480, (M,), dtype=torch.int32)], 1)
    vox = torch.zeros(M, P, Fd)
    for i in range(M):
        cx = coors[i, 3].item() * 0.25 - 60 + torch.rand(npts[i]) * 0.25
        cy = coors[i, 2].item() * 0.25 - 40 + torch.rand(npts[i]) * 0.25
        vox[i, :npts[i], 0], vox[i, :npts[i], 1] = cx, cy
        vox[i, :npts[i], 2:] = torch.randn(npts[i], Fd - 2)
    out["g6_voxels"], out["g6_num_points"], out["g6_coors"] = vox.clone().numpy(), npts.numpy(), coors.numpy()
    with torch.no_grad():
        y = net(vox.clone(), npts, coors)
    out["g6_pfn_out"] = y.numpy()
    out["g6_pfn_linear_w"] = net.pfn_layers[0].linear.weight.detach().numpy()
    out["g6_pfn_bn"] = np.stack([bn.weight.detach().numpy(), bn.bias.detach().numpy(),
                                 bn.running_mean.numpy(), bn.running_var.numpy()])
    # RadarPillarFeatureNet (RCFusion variant; 7 input dims, defect D8 avoided)
    rnet = pe.RadarPillarFeatureNet(in_channels=7, feat_channels=[64], with_distance=False, voxel_size=vsz,
                                    point_cloud_range=pcr, norm_cfg=dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01))
    rnet.eval()
    for m_ in rnet.modules():
        if isinstance(m_, nn.BatchNorm1d):
            m_.running_mean.normal_(); m_.running_var.uniform_(0.5, 2.0); m_.weight.data.uniform_(0.5, 1.5); m_.bias.data.normal_()
    vox7 = vox[:, :, :7].clone()
    with torch.no_grad():
        y7 = rnet(vox7.clone(), npts, coors)
    out["g6_radar_out"] = y7.numpy()
    sd = {k: v.numpy() for k, v in rnet.state_dict().items() if "num_batches" not in k}
    for k, v in sd.items():
        out["g6_radar_sd__" + k.replace(".", "__")] = v

    path = os.path.join(HERE, "reference_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    main()
