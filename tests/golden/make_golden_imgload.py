#!/usr/bin/env python3
"""Bookkeeping golden for ``LoadMultiViewImageFromFiles_newsc`` (projects/mmdet3d_plugin/datasets/pipelines/loading.py:
318-405) run on the reference class.  ``mmcv.imread`` / ``cv2.undistort`` / ``mmcv.imresize`` are absent from this
image; their stand-ins RECORD the arguments the reference hands them and return arrays of the right shape (imread:
a per-file constant image, undistort: its input, imresize: a constant image of the requested size), so what is pinned is
the reference's own logic — which views are halved, the float64 ``lidar2img`` / ``cam_intrinsic`` updates, the K and
distortion it passes to OpenCV, shapes, dtypes and keys.  Usage: python tests/golden/make_golden_imgload.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_data as G  # noqa: E402

CALLS = []
SIZES = {}


def _imread(name, flag="color"):
    CALLS.append(["imread", name.split("/")[-2], flag])
    h, w = SIZES[name]
    return np.full((h, w, 3), len(CALLS), dtype=np.uint8)


def _undistort(img, K, dist, R, newK):
    CALLS.append(["undistort", list(img.shape), np.asarray(K).tolist(), np.asarray(dist).tolist(), R is None,
                  np.asarray(newK).tolist()])
    return img


def _imresize(img, size, return_scale=False, interpolation="bilinear"):
    CALLS.append(["imresize", list(img.shape), [int(size[0]), int(size[1])], bool(return_scale)])
    return np.full((size[1], size[0]) + img.shape[2:], 7, dtype=img.dtype)


def main():
    G.install_stubs()
    mm, cv = sys.modules["mmcv"], sys.modules["cv2"]
    mm.imread, mm.imresize, cv.undistort = _imread, _imresize, _undistort
    sys.path.insert(0, G.REF)
    loading = G.load_file("ref_loading", "projects/mmdet3d_plugin/datasets/pipelines/loading.py")
    rng = np.random.default_rng(21)
    cams = ["camera_front", "camera_left_front", "camera_right_front", "camera_back", "camera_left_back", "camera_right_back"]
    names = [f"/data/cameras/{c}/{i:03d}.jpg" for i, c in enumerate(cams)]
    for n, c in zip(names, cams):
        SIZES[n] = (216, 384) if c in ("camera_front", "camera_back") else (108, 192)
    K = [np.eye(4) for _ in cams]
    for k in K:
        k[0, 0], k[1, 1], k[0, 2], k[1, 2] = rng.uniform(100, 300, 4)
    dist = [rng.normal(0, 0.05, 5) for _ in cams]
    l2i = [rng.normal(size=(4, 4)) for _ in cams]
    out = dict(names=names, sizes=[SIZES[n] for n in names], K=[k.tolist() for k in K], dist=[d.tolist() for d in dist],
               lidar2img=[m.tolist() for m in l2i], runs=[])
    for to_float32 in (False, True):
        CALLS.clear()
        res = loading.LoadMultiViewImageFromFiles_newsc(to_float32=to_float32)(
            dict(img_filename=list(names), cam_intrinsic=[k.copy() for k in K], cam_distortion=[d.copy() for d in dist],
                 lidar2img=[m.copy() for m in l2i]))
        out["runs"].append(dict(
            to_float32=to_float32, calls=list(CALLS), keys=sorted(res.keys()), dtype=str(res["img"][0].dtype),
            n_img=len(res["img"]), img0_shape=list(res["img"][0].shape), img_shape=list(res["img_shape"]),
            ori_shape=list(res["ori_shape"]), pad_shape=list(res["pad_shape"]), scale_factor=res["scale_factor"],
            norm=dict(mean=res["img_norm_cfg"]["mean"].tolist(), std=res["img_norm_cfg"]["std"].tolist(),
                      to_rgb=res["img_norm_cfg"]["to_rgb"], dtype=str(res["img_norm_cfg"]["mean"].dtype)),
            lidar2img=[np.asarray(m).tolist() for m in res["lidar2img"]],
            cam_intrinsic=[np.asarray(m).tolist() for m in res["cam_intrinsic"]],
            img_values=[int(v[0, 0, 0]) for v in res["img"]], filename_same=res["filename"] == names))
    path = os.path.join(HERE, "imgload_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
