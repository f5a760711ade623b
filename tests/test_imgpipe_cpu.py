"""Image-side pipeline steps against vectors captured from the reference classes
(tests/golden/make_golden_imgpipe.py -> imgpipe_golden.npz): the depth ground-truth format bit for bit,
the normalise / scale / pad / collect bookkeeping (shapes, keys, float64 lidar2img, the sizes the reference
asks mmcv for), and hand-computed cases for the restated mmcv/OpenCV pixel arithmetic (unpinned)."""
import json
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "imgpipe_golden.npz"))


def _depth_tree(gold, tmp_path):
    names = []
    for ci, cam in enumerate(gold["depth_cams"].tolist()):
        os.makedirs(tmp_path / "cameras" / cam, exist_ok=True)
        os.makedirs(tmp_path / "depth_gt" / cam, exist_ok=True)
        name = str(tmp_path / "cameras" / cam / f"{ci:04d}.jpg")
        gold[f"depth_in_{ci}"].tofile(name.replace("cameras", "depth_gt") + ".bin")
        names.append(name)
    return names


@pytest.mark.parametrize("tag,kw", [("half", dict(scale=0.5)), ("full", dict(scale=1.0)), ("half_pad8", dict(scale=0.5, pad=8))])
def test_load_gt_depth_is_bit_identical_to_the_reference(gold, tmp_path, tag, kw):
    from omnihd_amd.mm import PIPELINES
    from projects.mmdet3d_plugin.datasets.pipelines import LoadGTDepth
    assert PIPELINES.get("LoadGTDepth") is LoadGTDepth
    names = _depth_tree(gold, tmp_path)
    res = LoadGTDepth(**kw)(dict(filename=list(names)))
    d = res["img_depth"]
    assert isinstance(d, torch.Tensor) and d.dtype == torch.float32
    assert np.array_equal(d.numpy(), gold[f"depth_{tag}"])
    if tag == "half":          # the symmetric 2-row padding of the reference (540 -> 544)
        assert d.shape == (6, 544, 960) and float(d[:, :2].abs().sum()) == 0 and float(d[:, -2:].abs().sum()) == 0


def test_depth_map_feeds_the_gaussian_target(gold, tmp_path):
    """LoadGTDepth -> generate_guassian_depth_target: the two halves of the depth supervision fit together."""
    from projects.mmdet3d_plugin.datasets.pipelines import LoadGTDepth
    from projects.mmdet3d_plugin.utils.gaussian import generate_guassian_depth_target
    d = LoadGTDepth(scale=0.5)(dict(filename=_depth_tree(gold, tmp_path)))["img_depth"][None]       # (1, 6, 544, 960)
    target, min_depth = generate_guassian_depth_target(d, 4, [1, 60, 1], constant_std=0.5)
    assert target.shape == (6, 136, 240, 59) and min_depth.shape == (6, 136, 240)            # B*N flattened
    assert int((min_depth > 0).sum()) > 10000
    assert torch.isfinite(target).all() and float(target.sum(-1).max()) <= 1.0 + 1e-4


def test_normalise_scale_pad_collect_bookkeeping_matches_the_reference(gold):
    from omnihd_amd.mm import PIPELINES
    from projects.mmdet3d_plugin.datasets.pipelines import (CustomCollect3D, NormalizeMultiviewImage, PadMultiViewImage,
                                                             RandomScaleImageMultiViewImage)
    book = json.loads(str(gold["book_json"]))
    for n in ("CustomCollect3D", "NormalizeMultiviewImage", "PadMultiViewImage", "RandomScaleImageMultiViewImage",
              "LoadRadarPointsMultiSweeps", "LoadOccupancy_Newscenes"):
        assert n in PIPELINES, n
    rng = np.random.default_rng(1)
    l2i = [m.copy() for m in gold["pipe_lidar2img_in"]]
    results = dict(img=[rng.uniform(0, 255, (1080, 1920, 3)).astype(np.float32) for _ in range(6)], lidar2img=l2i,
                   filename=["f"] * 6, pts_filename="x.bin", sample_idx="tok", box_type_3d="LiDAR", points="PTS",
                   gt_bboxes_3d="BOX", scene_token="scene", can_bus=np.arange(18.0))
    results = NormalizeMultiviewImage(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)(results)
    assert results["img"][0].dtype == np.float32
    results = RandomScaleImageMultiViewImage(scales=[0.5])(results)
    assert [list(s) for s in results["img_shape"]] == book["after_scale"]["img_shape"]
    assert [list(s) for s in results["ori_shape"]] == book["after_scale"]["ori_shape"]
    # the sizes the reference asked mmcv.imresize for (w, h)
    asked = [c[2] for c in book["calls"] if c[0] == "imresize"]
    assert [[im.shape[1], im.shape[0]] for im in results["img"]] == asked
    results = PadMultiViewImage(size_divisor=32)(results)
    for k in ("img_shape", "ori_shape", "pad_shape"):
        assert [list(s) for s in results[k]] == book[k], k
    assert results["pad_fixed_size"] == book["pad_fixed_size"] and results["pad_size_divisor"] == book["pad_size_divisor"]
    assert np.array_equal(np.stack(results["lidar2img"]), gold["pipe_lidar2img_out"])              # float64, bit for bit
    cfg = results["img_norm_cfg"]
    assert cfg["mean"].tolist() == book["norm_cfg"]["mean"] and cfg["std"].tolist() == book["norm_cfg"]["std"]
    assert str(cfg["mean"].dtype) == book["norm_cfg"]["mean_dtype"] and cfg["to_rgb"] == book["norm_cfg"]["to_rgb"]
    assert float(np.abs(results["img"][0][540:]).sum()) == 0.0                                     # 4 padded rows, zeros
    data = CustomCollect3D(keys=["gt_bboxes_3d", "gt_labels_3d", "img", "points", "img_depth"])(results)
    assert sorted(data.keys()) == book["collect_keys"]
    assert sorted(k for k, v in data.items() if v is None) == book["collect_none"]
    assert sorted(data["img_metas"].data.keys()) == book["meta_keys"] and data["img_metas"].cpu_only == book["meta_cpu_only"]
    assert list(CustomCollect3D(keys=[]).meta_keys) == book["default_meta_keys"]
    # odd sizes, no matrix update, fixed-size padding
    r2 = dict(img=[rng.uniform(0, 1, (541, 961, 3)).astype(np.float32)], lidar2img=[np.eye(4)])
    r2 = RandomScaleImageMultiViewImage(scales=[0.3], scale_lidar2img=False)(r2)
    r2 = PadMultiViewImage(size=(200, 320), pad_val=7)(r2)
    odd = book["odd"]
    assert [list(s) for s in r2["img_shape"]] == odd["img_shape"] and [list(s) for s in r2["ori_shape"]] == odd["ori_shape"]
    assert np.array_equal(r2["lidar2img"][0], np.eye(4)) == odd["lidar2img_unchanged"]
    assert list(r2["pad_fixed_size"]) == odd["pad_fixed_size"] and r2["pad_size_divisor"] == odd["pad_size_divisor"]
    assert float(r2["img"][0][-1, -1, 0]) == 7.0
    with pytest.raises(AssertionError):
        RandomScaleImageMultiViewImage(scales=[0.5, 0.8])
    with pytest.raises(AssertionError):
        PadMultiViewImage()


def test_pixel_arithmetic_known_answers():
    from projects.mmdet3d_plugin.datasets.pipelines.transform_3d import imnormalize, impad, imresize_bilinear
    img = np.array([[[10, 20, 30], [40, 50, 60]], [[70, 80, 90], [100, 110, 120]]], dtype=np.uint8)     # BGR
    out = imnormalize(img, np.array([1, 2, 3], np.float32), np.array([2, 4, 8], np.float32), to_rgb=True)
    assert out.dtype == np.float32
    assert np.array_equal(out[0, 0], np.array([(30 - 1) / 2, (20 - 2) / 4, (10 - 3) / 8], np.float32))  # R, G, B
    assert np.array_equal(imnormalize(img, [0, 0, 0], [1, 1, 1], to_rgb=False)[1, 1], np.array([100, 110, 120], np.float32))
    assert img[0, 0, 0] == 10                                                                           # input untouched
    # scale 0.5 = mean of each 2x2 block, horizontal pass first, each pass rounded once
    x = np.random.default_rng(0).uniform(-3, 3, (6, 8, 3)).astype(np.float32)
    half = imresize_bilinear(x, (4, 3))
    h = x[:, 0::2] * np.float32(0.5) + x[:, 1::2] * np.float32(0.5)
    want = h[0::2] * np.float32(0.5) + h[1::2] * np.float32(0.5)
    assert half.shape == (3, 4, 3) and np.array_equal(half, want)
    assert np.array_equal(imresize_bilinear(x, (8, 6)), x)                                              # identity
    up = imresize_bilinear(np.array([[[0.0], [4.0]]], dtype=np.float32), (4, 1))[0, :, 0]               # 2 -> 4 pixels
    assert np.allclose(up, [0.0, 1.0, 3.0, 4.0])                                                        # clamped borders
    p = impad(np.ones((2, 3, 1), np.float32), (4, 4), pad_val=0)
    assert p.shape == (4, 4, 1) and p.sum() == 6 and p[2:].sum() == 0


@pytest.mark.parametrize("tag,kw", [("half", dict(scale=0.5)), ("full", dict(scale=1.0)), ("half_pad8", dict(scale=0.5, pad=8))])
def test_device_depth_scatter_is_bit_identical_to_the_reference(gold, tmp_path, tag, kw):
    """The scatter written for the GPU (device-agnostic torch), run here on CPU tensors."""
    from projects.mmdet3d_plugin.datasets.pipelines import LoadGTDepth
    res = LoadGTDepth(device="cpu", **kw)(dict(filename=_depth_tree(gold, tmp_path)))
    assert res["img_depth"].dtype == torch.float32 and np.array_equal(res["img_depth"].numpy(), gold[f"depth_{tag}"])


def test_device_image_pipeline_is_bit_identical_to_the_host_steps():
    from projects.mmdet3d_plugin.datasets.pipelines import (NormalizeMultiviewImage, PadMultiViewImage,
                                                             RandomScaleImageMultiViewImage)
    from projects.mmdet3d_plugin.datasets.pipelines.device_prep import DeviceImagePipeline
    rng = np.random.default_rng(3)
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    for (H, W), scale in [((270, 480), 0.5), ((135, 241), 0.5), ((96, 160), 1.0), ((120, 200), 0.3)]:
        views = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
        l2i = [rng.normal(size=(4, 4)) for _ in range(3)]
        r = dict(img=[v for v in views], lidar2img=[m.copy() for m in l2i])
        r = NormalizeMultiviewImage(mean=mean, std=std, to_rgb=True)(r)
        r = RandomScaleImageMultiViewImage(scales=[scale])(r)
        r = PadMultiViewImage(size_divisor=32)(r)
        want = np.stack([im.transpose(2, 0, 1) for im in r["img"]])
        got, got_l2i = DeviceImagePipeline(mean, std, True, scale, 32, device="cpu")(views, l2i)
        assert got.dtype == torch.float32 and got.shape == want.shape and got.is_contiguous()
        assert np.array_equal(got.numpy(), want), (H, W, scale)
        assert all(np.array_equal(a, b) for a, b in zip(got_l2i, r["lidar2img"]))


def test_image_loader_bookkeeping_matches_the_reference(tmp_path):
    """Which views are halved, the float64 matrix updates, the K / distortion handed to the undistortion, shapes and
    keys: pinned by tests/golden/make_golden_imgload.py (reference class over recording stand-ins for mmcv / cv2)."""
    from projects.mmdet3d_plugin.datasets.pipelines import LoadMultiViewImageFromFiles_newsc
    from projects.mmdet3d_plugin.datasets.pipelines import loading as L
    with open(os.path.join(os.path.dirname(__file__), "golden", "imgload_golden.json")) as f:
        g = json.load(f)
    sizes = dict(zip(g["names"], g["sizes"]))
    seen = []
    real_undistort = L.undistort

    def spy(img, K, dist):
        seen.append([list(img.shape), np.asarray(K).tolist(), np.asarray(dist).tolist()])
        return img                                               # identity, like the stand-in of the golden run
    L.undistort = spy
    try:
        for run in g["runs"]:
            seen.clear()
            reads = []

            def read(name):
                reads.append(name)
                return np.full(tuple(sizes[name]) + (3,), len(reads), dtype=np.uint8)
            res = LoadMultiViewImageFromFiles_newsc(to_float32=run["to_float32"], read=read)(
                dict(img_filename=list(g["names"]), cam_intrinsic=[np.array(k) for k in g["K"]],
                     cam_distortion=[np.array(d) for d in g["dist"]], lidar2img=[np.array(m) for m in g["lidar2img"]]))
            assert sorted(res.keys()) == run["keys"] and str(res["img"][0].dtype) == run["dtype"]
            assert len(res["img"]) == run["n_img"] and list(res["img"][0].shape) == run["img0_shape"]
            for k in ("img_shape", "ori_shape", "pad_shape"):
                assert list(res[k]) == run[k], k
            assert res["scale_factor"] == run["scale_factor"] and (res["filename"] == g["names"]) == run["filename_same"]
            n = res["img_norm_cfg"]
            assert n["mean"].tolist() == run["norm"]["mean"] and n["std"].tolist() == run["norm"]["std"]
            assert n["to_rgb"] == run["norm"]["to_rgb"] and str(n["mean"].dtype) == run["norm"]["dtype"]
            assert np.array_equal(np.array(res["lidar2img"]), np.array(run["lidar2img"]))            # float64, bit for bit
            assert np.array_equal(np.array(res["cam_intrinsic"]), np.array(run["cam_intrinsic"]))
            want_und = [[c[1], c[2], c[3]] for c in run["calls"] if c[0] == "undistort"]
            assert seen == want_und                                 # same image shapes, K[:3,:3] and distortion vectors
            assert all(c[4] is True and c[5] == c[2] for c in run["calls"] if c[0] == "undistort")   # R=None, newK=K
            halved = [c[2] for c in run["calls"] if c[0] == "imresize"]
            ours = [[im.shape[1], im.shape[0]] for im, name in zip(res["img"], g["names"])
                    if name.split("/")[-2] in ("camera_front", "camera_back")]
            assert ours == halved
    finally:
        L.undistort = real_undistort


def test_undistort_known_answers_and_decoder(tmp_path):
    from projects.mmdet3d_plugin.datasets.pipelines.loading import _read_image_bgr, undistort, undistort_map
    K = np.array([[200.0, 0, 96.0], [0, 210.0, 54.0], [0, 0, 1]])
    img = np.random.default_rng(0).integers(0, 256, (108, 192, 3), dtype=np.uint8)
    assert np.array_equal(undistort(img, K, np.zeros(5)), img)                    # no distortion: identity
    mx, my = undistort_map(K, [0.1, 0, 0, 0, 0], 108, 192)                         # pure k1
    u, v = 150, 20
    x, y = (u - 96.0) / 200.0, (v - 54.0) / 210.0
    r2 = x * x + y * y
    assert abs(mx[v, u] - (200.0 * x * (1 + 0.1 * r2) + 96.0)) < 1e-9 and abs(my[v, u] - (210.0 * y * (1 + 0.1 * r2) + 54.0)) < 1e-9
    assert mx[54, 96] == 96.0 and my[54, 96] == 54.0                               # the principal point stays
    mx, my = undistort_map(K, [0, 0, 0.01, -0.02], 108, 192)                       # tangential only, 4 coefficients
    assert abs(mx[v, u] - (200.0 * (x + 2 * 0.01 * x * y - 0.02 * (r2 + 2 * x * x)) + 96.0)) < 1e-9
    out = undistort(np.full((108, 192, 3), 200, np.uint8), K, [0.5, 0, 0, 0, 0])   # barrel: corners sample outside -> 0
    assert out[54, 96, 0] == 200 and out[0, 0, 0] == 0
    ramp = np.tile(np.arange(192, dtype=np.float32)[None, :, None], (108, 1, 1))   # linear image: bilinear is exact
    got = undistort(ramp, K, [0.05, 0, 0, 0, 0])
    mx, my = undistort_map(K, [0.05, 0, 0, 0, 0], 108, 192)
    inside = (mx >= 0) & (mx <= 191) & (my >= 0) & (my <= 107)
    assert np.allclose(got[..., 0][inside], mx[inside], atol=1e-3)
    with pytest.raises(ValueError):
        undistort_map(K, np.zeros(14), 4, 4)
    from PIL import Image                                                           # decoder: OpenCV channel order
    rgb = np.zeros((4, 5, 3), np.uint8); rgb[..., 0] = 250; rgb[..., 2] = 10
    Image.fromarray(rgb).save(tmp_path / "a.png")
    bgr = _read_image_bgr(str(tmp_path / "a.png"))
    assert bgr.shape == (4, 5, 3) and bgr[0, 0].tolist() == [10, 0, 250]


def test_device_image_loader_plus_pipeline_equal_the_host_chain():
    """decoded uint8 views -> undistort -> halve front/back -> normalise -> scale 0.5 -> pad: device classes (run on CPU
    tensors) against the host mirrors, bit for bit, including the matrices."""
    from projects.mmdet3d_plugin.datasets.pipelines import (LoadMultiViewImageFromFiles_newsc, NormalizeMultiviewImage,
                                                             PadMultiViewImage, RandomScaleImageMultiViewImage)
    from projects.mmdet3d_plugin.datasets.pipelines.device_prep import DeviceImageLoader, DeviceImagePipeline
    rng = np.random.default_rng(9)
    cams = ["camera_front", "camera_left_front", "camera_right_front", "camera_back", "camera_left_back", "camera_right_back"]
    names = [f"/d/cameras/{c}/0.jpg" for c in cams]
    big = lambda n: n.split("/")[-2] in ("camera_front", "camera_back")          # noqa: E731  (stored at twice the resolution)
    decoded = {n: rng.integers(0, 256, (216, 384, 3) if big(n) else (108, 192, 3), dtype=np.uint8) for n in names}
    K = []
    for n in names:
        h, w = decoded[n].shape[:2]
        k = np.eye(4); k[0, 0], k[1, 1], k[0, 2], k[1, 2] = 0.9 * w, 0.95 * w, w / 2 + 1.5, h / 2 - 2.0
        K.append(k)
    dist = [np.array([-0.12, 0.03, 1e-3, -2e-3, 0.01]) * (1 + 0.1 * i) for i in range(6)]
    l2i = [rng.normal(size=(4, 4)) for _ in range(6)]
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    r = LoadMultiViewImageFromFiles_newsc(to_float32=True, read=lambda n: decoded[n])(
        dict(img_filename=list(names), cam_intrinsic=[k.copy() for k in K], cam_distortion=dist, lidar2img=[m.copy() for m in l2i]))
    host_views = np.stack(r["img"])
    r = NormalizeMultiviewImage(mean=mean, std=std, to_rgb=True)(r)
    r = RandomScaleImageMultiViewImage(scales=[0.5])(r)
    r = PadMultiViewImage(size_divisor=32)(r)
    want = np.stack([im.transpose(2, 0, 1) for im in r["img"]])
    views, d_l2i, d_k = DeviceImageLoader(device="cpu")([decoded[n] for n in names], names, K, dist, l2i)
    assert views.dtype == torch.uint8 and np.array_equal(views.numpy().astype(np.float32), host_views)
    assert all(np.array_equal(a, b) for a, b in zip(d_k, r["cam_intrinsic"]))
    got, got_l2i = DeviceImagePipeline(mean, std, True, 0.5, 32, device="cpu")(views, d_l2i)
    assert got.shape == want.shape == (6, 3, 64, 96) and np.array_equal(got.numpy(), want)
    assert all(np.array_equal(a, b) for a, b in zip(got_l2i, r["lidar2img"]))
    plain = np.stack([decoded[n][::2, ::2] if big(n) else decoded[n] for n in names]).astype(np.float32)
    assert float(np.abs(host_views - plain).mean()) > 0.5                           # the undistortion did move pixels
