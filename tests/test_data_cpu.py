"""Data-format rows either side of the detector against golden vectors
produced by the reference's own Python (tests/golden/make_golden_data.py -> data_golden.npz):
radar sweep merge + ego-motion compensation, camera matrix composition, detector output ->
benchmark records; plus an end-to-end dataset.evaluate() on synthetic info records."""
import math
import os

import numpy as np
import pytest
import torch

from projects.mmdet3d_plugin.datasets import NewScenesDataset, camera_matrices, output_to_newsc_box
from projects.mmdet3d_plugin.datasets.pipelines.loading import (RADAR_ID, LoadRadarPointsMultiSweeps, RadarPoints,
                                                                half_scale_front_back, merge_radar_sweeps,
                                                                quaternion_rotation_matrix, scale_lidar2img)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = list(RADAR_ID)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "data_golden.npz"))


def _radars(gold, tmpdir):
    """Rebuild the sweep records + .bin files of the golden case from the stored arrays."""
    radars, off = {}, 0
    for row in gold["radar_meta"]:
        ri, si, n, ts = int(row[0]), int(row[1]), int(row[2]), int(row[3])
        pts = gold["radar_raw"][off:off + n]
        off += n
        path = os.path.join(str(tmpdir), f"r{ri}_{si}.bin")
        pts.astype(np.float32).tofile(path)
        radars.setdefault(NAMES[ri], []).append(dict(
            data_path=path, timestamp=ts, ego_velocity=row[4:7].tolist(), sensor2ego_rotation=row[7:11].tolist(),
            sensor2lidar_rotation=row[11:20].reshape(3, 3), sensor2lidar_translation=row[20:23]))
    return radars


def test_radar_merge_is_bit_identical_to_the_reference_loader(gold, tmp_path):
    radars = _radars(gold, tmp_path)
    got = merge_radar_sweeps(radars, LoadRadarPointsMultiSweeps._load_points, sweeps_num=3, load_dim=8)
    assert got.dtype == np.float64 and np.array_equal(got, gold["radar_points_all10"])
    assert np.array_equal(got[:, :8], gold["radar_points_use8"])
    assert set(np.unique(got[:, 9])) == {0.0, 1.0, 2.0, 3.0, 4.0, 5.0}
    assert got[:, 7].min() == 0.0 and 0.13 < got[:, 7].max() < 0.14        # dt of the third sweep: 2 x 66.667 ms


def test_radar_loader_class_filters_range_and_casts(gold, tmp_path):
    radars = _radars(gold, tmp_path)
    rng6 = [-60.0, -40.0, -3.0, 60.0, 40.0, 5.0]
    res = LoadRadarPointsMultiSweeps(load_dim=8, sweeps_num=3, use_dim=list(range(8)), max_num=40000, pc_range=rng6,
                                     file_client_args=dict(backend="disk"))({"radars": radars})
    pts = res["points"]
    want = torch.from_numpy(gold["radar_points_use8"]).float()
    m = ((want[:, 0] > -60) & (want[:, 1] > -40) & (want[:, 2] > -3) & (want[:, 0] < 60) & (want[:, 1] < 40) & (want[:, 2] < 5))
    assert isinstance(pts, RadarPoints) and pts.tensor.dtype == torch.float32 and pts.points_dim == 8
    assert 0 < len(pts) < len(want) and torch.equal(pts.tensor, want[m])
    edge = RadarPoints(np.array([[60.0, 0, 0], [-60.0, 0, 0], [59.999, 39.999, 4.999], [0, 40.0, 0], [0, 0, -3.0]]), points_dim=3)
    assert edge.in_range_3d(rng6).tolist() == [False, False, True, False, False]        # strict inequalities
    with pytest.raises(NotImplementedError):
        LoadRadarPointsMultiSweeps(file_client_args=dict(backend="petrel"))
    np.save(os.path.join(str(tmp_path), "a.npy"), np.arange(16, dtype=np.float32))
    assert LoadRadarPointsMultiSweeps._load_points(os.path.join(str(tmp_path), "a.npy")).shape == (16,)


def test_quaternion_rotation_matrix_known_answers():
    np.testing.assert_allclose(quaternion_rotation_matrix((math.cos(0.25), 0, 0, math.sin(0.25))),
                               [[math.cos(0.5), -math.sin(0.5), 0], [math.sin(0.5), math.cos(0.5), 0], [0, 0, 1]], atol=1e-15)
    r = quaternion_rotation_matrix((2.0, 0.2, -0.4, 1.0))                   # un-normalised
    np.testing.assert_allclose(r @ r.T, np.eye(3), atol=1e-14)
    assert abs(np.linalg.det(r) - 1) < 1e-14


def test_camera_matrices_match_reference(gold):
    for i, row in enumerate(gold["cam_inputs"]):
        info = dict(sensor2lidar_rotation=row[:9].reshape(3, 3), sensor2lidar_translation=row[9:12],
                    cam_intrinsic=row[12:21].reshape(3, 3), cam_distortion=row[21:26])
        l2i, k, l2c = camera_matrices(info)
        assert np.array_equal(l2i, gold["cam_lidar2img"][i]) and np.array_equal(k, gold["cam_intrinsic"][i])
        assert np.array_equal(l2c, gold["cam_lidar2cam"][i])
        # a point 10 m in front of the camera projects to the principal point
        centre = info["sensor2lidar_rotation"] @ np.array([0, 0, 10.0]) + info["sensor2lidar_translation"]
        uvw = l2i @ np.append(centre, 1.0)
        np.testing.assert_allclose(uvw[:2] / uvw[2], [k[0, 2], k[1, 2]], atol=1e-6)
    names = ["/d/camera_front/a.jpg", "/d/camera_left_front/a.jpg", "/d/camera_back/a.jpg"]
    l2i, k = half_scale_front_back(names, list(gold["cam_lidar2img"][:3]), list(gold["cam_intrinsic"][:3]))
    assert np.array_equal(l2i[0][:2], 0.5 * gold["cam_lidar2img"][0][:2]) and np.array_equal(l2i[1], gold["cam_lidar2img"][1])
    assert np.array_equal(k[2][:2], 0.5 * gold["cam_intrinsic"][2][:2]) and np.array_equal(l2i[0][2:], gold["cam_lidar2img"][0][2:])
    assert np.array_equal(scale_lidar2img(l2i, 0.5)[1][:2], 0.5 * l2i[1][:2])


def test_output_to_newsc_box_matches_reference(gold):
    from newscenes_devkit.eval.detection.config import config_factory
    from omnihd_amd.mm.boxes import LiDARInstance3DBoxes
    det = dict(boxes_3d=LiDARInstance3DBoxes(torch.from_numpy(gold["o2n_boxes"]), box_dim=9),
               scores_3d=torch.from_numpy(gold["o2n_scores"]), labels_3d=torch.from_numpy(gold["o2n_labels"]))
    got = output_to_newsc_box(det, NewScenesDataset.CLASSES, config_factory("detection_newsc_config_final"))
    assert len(got) == len(gold["o2n_score"]) < len(gold["o2n_boxes"])
    assert np.array_equal(np.array([b["center"] for b in got]), gold["o2n_center"])
    assert np.array_equal(np.array([b["wlh"] for b in got]), gold["o2n_wlh"])
    np.testing.assert_allclose(np.array([b["orientation"] for b in got]), gold["o2n_quat"], rtol=0, atol=1e-15)
    assert np.array_equal(np.array([b["velocity"] for b in got]), gold["o2n_velocity"])
    assert np.array_equal(np.array([b["score"] for b in got]), gold["o2n_score"].astype(np.float64))
    assert [b["label"] for b in got] == gold["o2n_label"].tolist()


def _infos(rng, n_samples=6, n_boxes=8):
    infos = []
    for s in range(n_samples):
        names = rng.choice(["car", "pedestrian", "rider", "large_vehicle"], n_boxes)
        boxes = np.concatenate([rng.uniform(-55, 55, (n_boxes, 1)), rng.uniform(-35, 35, (n_boxes, 1)), rng.uniform(-1, 1, (n_boxes, 1)),
                                rng.uniform(0.5, 5, (n_boxes, 3)), rng.uniform(-3, 3, (n_boxes, 1))], 1)
        vel = rng.normal(0, 3, (n_boxes, 2))
        vel[0] = np.nan
        infos.append(dict(token=f"s{s}", timestamp=1000 - s, lidar_path="", sweeps=[], radars={}, cams={}, gt_boxes=boxes,
                          gt_names=names, gt_velocity=vel, valid_flag=np.ones(n_boxes, dtype=bool)))
    return infos


def test_dataset_evaluate_end_to_end_with_perfect_and_shifted_detections():
    from omnihd_amd.mm.boxes import LiDARInstance3DBoxes
    rng = np.random.default_rng(5)
    ds = NewScenesDataset(data_infos=_infos(rng), test_mode=True)
    assert [i["token"] for i in ds.data_infos] == ["s5", "s4", "s3", "s2", "s1", "s0"]     # sorted by timestamp
    ann = ds.get_ann_info(0)
    assert ann["gt_bboxes_3d"].tensor.shape == (8, 9) and float(ann["gt_bboxes_3d"].tensor[0, 7]) == 0.0      # NaN velocity -> 0
    np.testing.assert_allclose(ann["gt_bboxes_3d"].gravity_center.numpy(), ds.data_infos[0]["gt_boxes"][:, :3], atol=1e-5)

    def results(shift):
        out = []
        for i in range(len(ds)):
            a = ds.get_ann_info(i)
            t = a["gt_bboxes_3d"].tensor.clone()
            t[:, 0] += shift
            out.append(dict(pts_bbox=dict(boxes_3d=LiDARInstance3DBoxes(t, box_dim=9), scores_3d=torch.linspace(0.9, 0.3, len(t)),
                                          labels_3d=torch.from_numpy(a["gt_labels_3d"]))))
        return out
    perfect = ds.evaluate(results(0.0))
    assert perfect["pts_bbox_NewScenes/mAP"] == pytest.approx(1.0) and perfect["pts_bbox_NewScenes/mATE"] < 1e-4
    assert perfect["pts_bbox_NewScenes/mAOE"] < 1e-4 and perfect["pts_bbox_NewScenes/mASE"] < 1e-4
    assert perfect["pts_bbox_NewScenes/NOS"] == pytest.approx(1.0, abs=1e-4)
    assert "pts_bbox_NewScenes/car_AP_dist_1.0" in perfect and "pts_bbox_NewScenes/rider_trans_err" in perfect
    shifted = ds.evaluate(results(1.5))                      # 1.5 m off: missed at the 1 m threshold, hit at 2/3/4 m
    assert shifted["pts_bbox_NewScenes/car_AP_dist_1.0"] < 0.2 and shifted["pts_bbox_NewScenes/car_AP_dist_2.0"] > 0.8
    assert 1.4 < shifted["pts_bbox_NewScenes/mATE"] < 1.6
    sub = ds._format_bbox([r["pts_bbox"] for r in results(0.0)])
    assert set(sub) == {"meta", "results"} and set(sub["results"]) == {f"s{i}" for i in range(6)}
    rec = sub["results"]["s5"][0]
    assert set(rec) == {"sample_token", "translation", "size", "rotation", "velocity", "detection_name", "detection_score"}


def test_radar_points_augmentation_hooks_match_the_reference():
    """flip / scale / rotate / in_range_bev: positions AND the compensated velocity columns move together
    (tests/golden/make_golden_points.py ran the reference class)."""
    import os
    from projects.mmdet3d_plugin.core.points.radar_points import RadarPoints
    from projects.mmdet3d_plugin.datasets.pipelines import RadarPoints as SamePoints
    assert RadarPoints is SamePoints
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "points_golden.npz"))
    new = lambda: RadarPoints(g["pts"].copy(), points_dim=8)          # noqa: E731
    for d in ("horizontal", "vertical"):
        p = new(); p.flip(d)
        assert np.array_equal(p.tensor.numpy(), g[f"flip_{d}"]), d
    p = new(); p.scale(1.25)
    assert np.array_equal(p.tensor.numpy(), g["scale"])
    for tag, (rot, axis) in {"z": (0.3, None), "y": (-0.7, 1), "x": (1.1, 0), "m1": (0.5, -1)}.items():
        p = new()
        T = p.rotate(rot, axis)
        assert np.array_equal(np.asarray(T), g[f"rot_{tag}_T"]) and np.array_equal(p.tensor.numpy(), g[f"rot_{tag}"]), tag
    p = new()
    T = p.rotate(torch.from_numpy(g["rot_mat_in"]))
    assert np.array_equal(np.asarray(T), g["rot_mat_T"]) and np.array_equal(p.tensor.numpy(), g["rot_mat"])
    assert np.array_equal(new().in_range_bev(g["bev_range"].tolist()).numpy(), g["in_bev"])
    with pytest.raises(ValueError):
        new().rotate(0.1, axis=5)


def test_reference_module_paths_of_head_and_points_exist():
    from omnihd_amd.mm import HEADS
    from projects.mmdet3d_plugin.bevfusion.dense_heads.det_anchor3d_head import Anchor3DHeadV1
    assert HEADS.get("Anchor3DHeadV1") is Anchor3DHeadV1
