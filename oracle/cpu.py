"""oracle/cpu.py — TEST INFRASTRUCTURE, NOT PRODUCT.

numpy-facing wrappers of the C oracle (oracle/liboracle.so, built by oracle/Makefile).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_hard_voxelize.restype = ctypes.c_int
        _lib.oracle_nms_rotated.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def bev_pool_v2_fwd(depth, feat, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape, starts, lengths,
                    threads=False):
    """-> out (B,Z,Y,X,C) fp32, zeros where no interval writes (reference: bev_pool.py:27)."""
    depth, feat = _f32(depth), _f32(feat)
    rd, rf, rb, st, ln = map(_i32, (ranks_depth, ranks_feat, ranks_bev, starts, lengths))
    out = np.zeros(tuple(int(s) for s in bev_feat_shape), dtype=np.float32)
    fn = lib().oracle_bev_pool_v2_fwd_omp if threads else lib().oracle_bev_pool_v2_fwd
    fn(ctypes.c_int(feat.shape[-1]), ctypes.c_int(len(st)), _p(depth), _p(feat), _p(rd), _p(rf), _p(rb),
       _p(st), _p(ln), _p(out))
    return out


def bev_pool_v2_bwd(out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev, starts, lengths, threads=False):
    """Backward tables in, (depth_grad, feat_grad) out (zeros where untouched, bev_pool.py:67-68)."""
    og, depth, feat = _f32(out_grad), _f32(depth), _f32(feat)
    rd, rf, rb, st, ln = map(_i32, (ranks_depth, ranks_feat, ranks_bev, starts, lengths))
    dg, fg = np.zeros_like(depth), np.zeros_like(feat)
    fn = lib().oracle_bev_pool_v2_bwd_omp if threads else lib().oracle_bev_pool_v2_bwd
    fn(ctypes.c_int(feat.shape[-1]), ctypes.c_int(len(st)), _p(og), _p(depth), _p(feat), _p(rd), _p(rf),
       _p(rb), _p(st), _p(ln), _p(dg), _p(fg))
    return dg, fg


def bev_pool_v1_fwd(x, geom_feats, starts, lengths, b, d, h, w):
    x, g, st, ln = _f32(x), _i32(geom_feats), _i32(starts), _i32(lengths)
    n, c = x.shape
    out = np.zeros((b, d, h, w, c), dtype=np.float32)
    lib().oracle_bev_pool_v1_fwd(*[ctypes.c_int(v) for v in (b, d, h, w, n, c, len(st))], _p(x), _p(g),
                                 _p(st), _p(ln), _p(out))
    return out


def bev_pool_v1_bwd(out_grad, geom_feats, starts, lengths, b, d, h, w):
    og, g, st, ln = _f32(out_grad), _i32(geom_feats), _i32(starts), _i32(lengths)
    n, c = g.shape[0], og.shape[-1]
    xg = np.zeros((n, c), dtype=np.float32)
    lib().oracle_bev_pool_v1_bwd(*[ctypes.c_int(v) for v in (b, d, h, w, n, c, len(st))], _p(og), _p(g),
                                 _p(st), _p(ln), _p(xg))
    return xg


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    pts = _f32(points)
    n, f = pts.shape
    vs, cr = _f32(voxel_size), _f32(coors_range)
    voxels = np.zeros((max_voxels, max_points, f), dtype=np.float32)
    coors = np.zeros((max_voxels, 3), dtype=np.int32)
    num = np.zeros((max_voxels,), dtype=np.int32)
    m = lib().oracle_hard_voxelize(_p(pts), ctypes.c_int(n), ctypes.c_int(f), _p(vs), _p(cr),
                                   ctypes.c_int(max_points), ctypes.c_int(max_voxels), _p(voxels), _p(coors), _p(num))
    return voxels[:m], coors[:m], num[:m]


def pillar_scatter(feats, coors, batch, ny, nx):
    feats, coors = _f32(feats), _i32(coors)
    m, c = feats.shape
    canvas = np.empty((batch, c, ny, nx), dtype=np.float32)
    lib().oracle_pillar_scatter(_p(feats), _p(coors), ctypes.c_int(m), ctypes.c_int(c), ctypes.c_int(batch),
                                ctypes.c_int(ny), ctypes.c_int(nx), _p(canvas))
    return canvas


def iou_bev_matrix(boxes_a, boxes_b):
    """(Na,5),(Nb,5) boxes (x1,y1,x2,y2,ry) -> (Na,Nb) rotated BEV IoU (oracle/nms_oracle.c)."""
    a, b = _f32(boxes_a).reshape(-1, 5), _f32(boxes_b).reshape(-1, 5)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    lib().oracle_iou_bev_matrix(_p(a), ctypes.c_int(a.shape[0]), _p(b), ctypes.c_int(b.shape[0]), _p(out))
    return out


def nms_rotated_sorted(sorted_boxes, thresh):
    """Boxes already in descending score order -> kept positions (int64, ascending)."""
    b = _f32(sorted_boxes).reshape(-1, 5)
    keep = np.zeros(max(b.shape[0], 1), dtype=np.int64)
    n = lib().oracle_nms_rotated(_p(b), ctypes.c_int(b.shape[0]), ctypes.c_float(thresh), _p(keep))
    return keep[:n]


def nms_rotated(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """mmdet3d `nms_gpu` semantics on numpy arrays; the score sort is a STABLE descending sort."""
    order = np.argsort(-np.asarray(scores, dtype=np.float32), kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    kept = order[nms_rotated_sorted(np.asarray(boxes)[order], thresh)]
    return kept if post_max_size is None else kept[:post_max_size]


def deform_conv(x, offset, weight, stride=1, pad=1, dil=1, groups=1, deform_groups=1, grad_out=None):
    """mmcv 1.4.0 deformable convolution v1 (oracle/dcn_oracle.c): x (B,C,H,W), offset (B,DG*2*K*K,Ho,Wo), weight
    (N,C/G,K,K) -> out (B,N,Ho,Wo); with ``grad_out`` also (grad_x, grad_offset, grad_weight)."""
    x, offset, weight = _f32(x), _f32(offset), _f32(weight)
    B, C, H, W = x.shape
    N, _, K, _ = weight.shape
    Ho, Wo = offset.shape[2:]
    out = np.zeros((B, N, Ho, Wo), dtype=np.float32)
    geo = [ctypes.c_int(v) for v in (B, C, H, W, N, K, stride, pad, dil, groups, deform_groups, Ho, Wo)]
    lib().oracle_deform_conv_fwd(_p(x), _p(offset), _p(weight), _p(out), *geo)
    if grad_out is None:
        return out
    g = _f32(grad_out)
    gx, go, gw = np.zeros_like(x), np.zeros_like(offset), np.zeros_like(weight)
    lib().oracle_deform_conv_bwd(_p(x), _p(offset), _p(weight), _p(g), _p(gx), _p(go), _p(gw), *geo)
    return out, gx, go, gw
