#!/usr/bin/env python3
"""Golden vectors for the image-side pipeline steps (depth ground-truth format and the
normalise / scale / pad / collect bookkeeping), produced by running the REFERENCE's own classes:

  * ``LoadGTDepth.__call__`` (projects/mmdet3d_plugin/datasets/pipelines/loading.py:17-63) on synthetic
    ``[u, v, d]`` float32 files in a temp tree — self-contained numpy: the depth maps are pinned bit for bit;
  * ``NormalizeMultiviewImage`` / ``RandomScaleImageMultiViewImage`` / ``PadMultiViewImage`` /
    ``CustomCollect3D`` (pipelines/transform_3d.py) chained as in bevfusion.py:180-189.  Their pixel
    arithmetic lives in mmcv/OpenCV, absent from this image; the ``mmcv`` stand-in below RECORDS the calls
    (function, output size / divisor / mean / std / to_rgb) and returns arrays of the right shape, so what
    is pinned is the reference's bookkeeping — shapes, keys, the float64 ``lidar2img`` update and the
    arguments it hands to mmcv — not pixel values.

Runs only in the authoring container.  Usage: python tests/golden/make_golden_imgpipe.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_data as G  # noqa: E402

CALLS = []


def _impad(img, shape=None, pad_val=0):
    CALLS.append(["impad", list(img.shape), list(shape), float(pad_val)])
    out = np.full(tuple(shape[:2]) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:img.shape[0], :img.shape[1]] = img
    return out


def _impad_to_multiple(img, divisor, pad_val=0):
    CALLS.append(["impad_to_multiple", list(img.shape), int(divisor), float(pad_val)])
    h = int(np.ceil(img.shape[0] / divisor)) * divisor
    w = int(np.ceil(img.shape[1] / divisor)) * divisor
    out = np.full((h, w) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:img.shape[0], :img.shape[1]] = img
    return out


def _imnormalize(img, mean, std, to_rgb=True):
    CALLS.append(["imnormalize", list(img.shape), str(img.dtype), np.asarray(mean).tolist(), np.asarray(std).tolist(),
                  bool(to_rgb)])
    return np.zeros(img.shape, dtype=np.float32)


def _imresize(img, size, return_scale=False, interpolation="bilinear"):
    CALLS.append(["imresize", list(img.shape), [int(size[0]), int(size[1])], bool(return_scale), interpolation])
    return np.zeros((size[1], size[0]) + img.shape[2:], dtype=img.dtype)


class _DC:
    def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
        self.data, self.cpu_only, self.stack = data, cpu_only, stack


def main():
    G.install_stubs()
    mm = sys.modules["mmcv"]
    mm.impad, mm.impad_to_multiple, mm.imnormalize, mm.imresize = _impad, _impad_to_multiple, _imnormalize, _imresize
    G._mod("mmcv.parallel", DataContainer=_DC)
    sys.path.insert(0, G.REF)
    loading = G.load_file("ref_loading", "projects/mmdet3d_plugin/datasets/pipelines/loading.py")
    t3d = G.load_file("ref_transform_3d", "projects/mmdet3d_plugin/datasets/pipelines/transform_3d.py")
    rng = np.random.default_rng(77)
    out = {}

    # ---------------- LoadGTDepth ----------------
    cams = ["camera_front", "camera_left_front", "camera_right_front", "camera_back", "camera_left_back", "camera_right_back"]
    tmp = tempfile.mkdtemp(prefix="depth_golden_")
    names = []
    for ci, cam in enumerate(cams):
        os.makedirs(os.path.join(tmp, "cameras", cam), exist_ok=True)
        os.makedirs(os.path.join(tmp, "depth_gt", cam), exist_ok=True)
        big = cam in ("camera_front", "camera_back")
        W, H = (3840, 2160) if big else (1920, 1080)
        n = 3000 + 100 * ci
        uvd = np.empty((n, 3), dtype=np.float32)
        uvd[:, 0] = rng.uniform(-20, W + 20, n)
        uvd[:, 1] = rng.uniform(-20, H + 20, n)
        uvd[:, 2] = rng.uniform(0.5, 80.0, n)
        uvd[:50, :2] = uvd[50:100, :2]                      # duplicates of the same pixel: the later row wins
        uvd[100, :2] = (0.0, 0.0)
        uvd[101, :2] = (W - 0.01, H - 0.01)                 # last pixel
        uvd[102, :2] = (float(W), float(H))                 # first pixel outside
        uvd[103, :2] = (-0.9, -0.9)                         # truncates to (0, 0) after scaling: inside
        name = os.path.join(tmp, "cameras", cam, f"{ci:04d}.jpg")
        uvd.tofile(name.replace("cameras", "depth_gt") + ".bin")
        names.append(name)
        out[f"depth_in_{ci}"] = uvd
    out["depth_cams"] = np.array(cams)
    for tag, kw in [("half", dict(scale=0.5)), ("full", dict(scale=1.0)), ("half_pad8", dict(scale=0.5, pad=8))]:
        res = loading.LoadGTDepth(**kw)(dict(filename=list(names)))
        out[f"depth_{tag}"] = res["img_depth"].numpy()
        print("LoadGTDepth", tag, tuple(res["img_depth"].shape), "non-zero", int((res["img_depth"] > 0).sum()))

    # ---------------- normalise -> scale -> pad -> collect (bookkeeping) ----------------
    l2i = [np.asarray(rng.normal(size=(4, 4))) for _ in range(6)]
    results = dict(img=[np.zeros((1080, 1920, 3), dtype=np.float32) for _ in range(6)], lidar2img=[m.copy() for m in l2i],
                   filename=names, pts_filename="x.bin", sample_idx="tok", box_type_3d="LiDAR", points="PTS",
                   gt_bboxes_3d="BOX", scene_token="scene", can_bus=np.arange(18.0))
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    np.random.seed(0)
    results = t3d.NormalizeMultiviewImage(mean=mean, std=std, to_rgb=True)(results)
    results = t3d.RandomScaleImageMultiViewImage(scales=[0.5])(results)
    after_scale = dict(img_shape=[list(s) for s in results["img_shape"]], ori_shape=[list(s) for s in results["ori_shape"]])
    results = t3d.PadMultiViewImage(size_divisor=32)(results)
    out["pipe_lidar2img_in"], out["pipe_lidar2img_out"] = np.stack(l2i), np.stack(results["lidar2img"])
    book = dict(after_scale=after_scale, img_shape=[list(s) for s in results["img_shape"]],
                ori_shape=[list(s) for s in results["ori_shape"]], pad_shape=[list(s) for s in results["pad_shape"]],
                pad_fixed_size=results["pad_fixed_size"], pad_size_divisor=results["pad_size_divisor"],
                norm_cfg=dict(mean=results["img_norm_cfg"]["mean"].tolist(), std=results["img_norm_cfg"]["std"].tolist(),
                              mean_dtype=str(results["img_norm_cfg"]["mean"].dtype), to_rgb=results["img_norm_cfg"]["to_rgb"]),
                calls=list(CALLS))
    data = t3d.CustomCollect3D(keys=["gt_bboxes_3d", "gt_labels_3d", "img", "points", "img_depth"])(results)
    book["collect_keys"] = sorted(data.keys())
    book["collect_none"] = sorted(k for k, v in data.items() if v is None)
    book["meta_keys"] = sorted(data["img_metas"].data.keys())
    book["meta_cpu_only"] = bool(data["img_metas"].cpu_only)
    book["default_meta_keys"] = list(t3d.CustomCollect3D(keys=[]).meta_keys)
    # fixed-size padding and an odd size through the scaler
    CALLS.clear()
    r2 = dict(img=[np.zeros((541, 961, 3), dtype=np.float32)], lidar2img=[np.eye(4)])
    r2 = t3d.RandomScaleImageMultiViewImage(scales=[0.3], scale_lidar2img=False)(r2)
    r2 = t3d.PadMultiViewImage(size=(200, 320), pad_val=7)(r2)
    book["odd"] = dict(img_shape=[list(s) for s in r2["img_shape"]], ori_shape=[list(s) for s in r2["ori_shape"]],
                       lidar2img_unchanged=bool(np.array_equal(r2["lidar2img"][0], np.eye(4))), calls=list(CALLS),
                       pad_fixed_size=list(r2["pad_fixed_size"]), pad_size_divisor=r2["pad_size_divisor"])
    out["book_json"] = np.array(json.dumps(book))
    path = os.path.join(HERE, "imgpipe_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    main()
