#!/usr/bin/env python3
"""Golden vectors for the data-format rows, produced by running the
REFERENCE's own Python on synthetic inputs:

  * ``LoadRadarPointsMultiSweeps.__call__`` (projects/mmdet3d_plugin/datasets/pipelines/loading.py:
    113-316) on synthetic 8-column float32 sweeps written to a temp directory;
  * ``NewScenesDataset.get_data_info`` (datasets/newscenes_dataset.py:164-234): lidar2img /
    cam_intrinsic / lidar2cam composition;
  * ``LoadMultiViewImageFromFiles_newsc`` matrix bookkeeping is NOT run (needs OpenCV images);
  * ``output_to_newsc_box`` (newscenes_dataset.py:537-583) with the reference devkit's ``Box``.

Absent third-party packages get inert stand-ins (registries, decorators, base classes); the two
stand-ins that carry arithmetic are documented where they are defined: ``Quaternion`` (textbook
rotation matrix / axis-angle constructor) and ``_Recorder`` (captures the float64 point array the
reference hands to ``RadarPoints`` *before* the upstream range filter, so everything stored was
computed by reference code).  Runs only in the authoring container.
Usage: python tests/golden/make_golden_data.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_eval import Quaternion  # noqa: E402

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Registry:
    def register_module(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda cls: cls


class _FileClient:
    def __init__(self, **kw):
        pass

    def get(self, path):
        with open(path, "rb") as f:
            return f.read()


class _Recorder:
    """Stands in for RadarPoints: keeps what the reference passes in; the range filter is a no-op."""
    last = None

    def __init__(self, tensor, points_dim=3, attribute_dims=None):
        _Recorder.last = np.array(tensor, copy=True)
        self.n = len(tensor)

    def in_range_3d(self, rng):
        return np.ones(self.n, dtype=bool)

    def __getitem__(self, item):
        return self


class _Boxes:
    """What ``output_to_newsc_box`` reads from a LiDARInstance3DBoxes: tensors we supply as inputs."""

    def __init__(self, t):
        self.tensor = t

    gravity_center = property(lambda s: torch.cat([s.tensor[:, :2], (s.tensor[:, 2] + s.tensor[:, 5] * 0.5)[:, None]], 1))
    dims = property(lambda s: s.tensor[:, 3:6])
    yaw = property(lambda s: s.tensor[:, 6])

    def __len__(self):
        return self.tensor.shape[0]


def install_stubs():
    pq = _mod("pyquaternion", Quaternion=Quaternion)
    pq.Quaternion = Quaternion
    _mod("mmcv", FileClient=_FileClient, check_file_exist=lambda p: None, track_iter_progress=lambda x: x)
    _mod("mmdet")
    _mod("mmdet.datasets", DATASETS=_Registry())
    _mod("mmdet.datasets.builder", PIPELINES=_Registry())
    _mod("cv2")
    _mod("mmdet3d")
    _mod("mmdet3d.core", show_result=None)
    _mod("mmdet3d.core.bbox", Box3DMode=None, Coord3DMode=None, LiDARInstance3DBoxes=None)
    _mod("mmdet3d.core.points", BasePoints=object, get_points_type=None)
    _mod("mmdet3d.datasets")
    _mod("mmdet3d.datasets.custom_3d", Custom3DDataset=object)
    _mod("mmdet3d.datasets.pipelines", Compose=None)
    for name in ("nuscenes", "nuscenes.eval", "nuscenes.eval.common", "nuscenes.utils"):
        _mod(name)
    _mod("nuscenes.eval.common.data_classes", EvalBox=object)
    _mod("nuscenes.utils.data_classes", Box=object)
    _mod("projects")
    _mod("projects.mmdet3d_plugin")
    _mod("projects.mmdet3d_plugin.core")
    _mod("projects.mmdet3d_plugin.core.points")
    _mod("projects.mmdet3d_plugin.core.points.radar_points", RadarPoints=_Recorder)
    _mod("projects.mmdet3d_plugin.core.vis_tools", project_pts_on_img=None)


def load_file(modname, relpath):
    import importlib.util
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def main():
    install_stubs()
    sys.path.insert(0, REF)
    out = {}
    rng = np.random.default_rng(2024)

    # ---------------- radar loader ----------------
    loading = load_file("ref_loading", "projects/mmdet3d_plugin/datasets/pipelines/loading.py")
    names = ["radar_front", "radar_left_front", "radar_right_front", "radar_back", "radar_left_back", "radar_right_back"]
    mount_yaw = [0.0, 1.0, -1.0, np.pi, 2.2, -2.2]
    tmp = tempfile.mkdtemp(prefix="radar_golden_")
    radars, raw, meta = {}, [], []
    for ri, name in enumerate(names):
        n_sweeps = 2 if name == "radar_back" else 4          # fewer sweeps than sweeps_num for one radar
        sweeps = []
        for si in range(n_sweeps):
            n = int(rng.integers(40, 120)) if not (ri == 4 and si == 1) else 0     # one empty sweep
            pts = np.zeros((n, 8), np.float32)
            pts[:, 0] = rng.uniform(1, 80, n)
            pts[:, 1] = rng.uniform(-40, 40, n)
            pts[:, 2] = rng.uniform(-4, 6, n)
            pts[:, 3] = rng.normal(0, 6, n)
            pts[:, 4] = rng.uniform(0, 60, n)
            pts[:, 5] = rng.integers(0, 3, n)
            pts[:, 6] = rng.uniform(0, 30, n)
            pts[:, 7] = 1
            path = os.path.join(tmp, f"{name}_{si}.bin")
            pts.tofile(path)
            yaw = mount_yaw[ri] + rng.normal(0, 0.01)
            q = np.array([np.cos(yaw / 2), 0.003, -0.002, np.sin(yaw / 2)])     # slightly off-unit: exercises normalisation
            R = rot_z(yaw + 0.002 * si) @ np.array([[1, 0, 0], [0, np.cos(0.01), -np.sin(0.01)], [0, np.sin(0.01), np.cos(0.01)]])
            t = np.array([np.cos(mount_yaw[ri]) * 2.0, np.sin(mount_yaw[ri]) * 0.9, 0.6]) + rng.normal(0, 0.3, 3) * si
            sweep = dict(data_path=path, timestamp=1700000000000000 + ri * 1000 - si * 66667,
                         ego_velocity=[float(rng.normal(8, 3)), float(rng.normal(0, 0.5)), float(rng.normal(0, 0.1))],
                         sensor2ego_rotation=q.tolist(), sensor2lidar_rotation=R, sensor2lidar_translation=t)
            sweeps.append(sweep)
            raw.append(pts)
            meta.append(np.concatenate([[ri, si, n, sweep["timestamp"]], sweep["ego_velocity"], q, R.ravel(), t]))
        radars[name] = sweeps
    loader = loading.LoadRadarPointsMultiSweeps(load_dim=8, sweeps_num=3, use_dim=[0, 1, 2, 3, 4, 5, 6, 7], max_num=40000,
                                                pc_range=[-60.0, -40.0, -3.0, 60.0, 40.0, 5.0])
    loader({"radars": radars})
    out["radar_raw"] = np.concatenate(raw, 0)
    out["radar_meta"] = np.array(meta, dtype=np.float64)            # per sweep: radar, sweep, n, ts, ego v(3), q(4), R(9), t(3)
    out["radar_points_use8"] = _Recorder.last                       # (M, 8) float64 after use_dim, before the range filter
    loader10 = loading.LoadRadarPointsMultiSweeps(load_dim=8, sweeps_num=3, use_dim=list(range(10)), pc_range=[-60.0, -40.0, -3.0, 60.0, 40.0, 5.0])
    loader10({"radars": radars})
    out["radar_points_all10"] = _Recorder.last
    print("radar:", out["radar_raw"].shape, "->", out["radar_points_all10"].shape)

    # ---------------- dataset: camera matrices + result conversion ----------------
    ds = load_file("ref_dataset", "projects/mmdet3d_plugin/datasets/newscenes_dataset.py")
    cams = {}
    cam_in = []
    for ci, cname in enumerate(["camera_front", "camera_left_front", "camera_right_front", "camera_back", "camera_left_back",
                                "camera_right_back"]):
        yaw = [0, 1.05, -1.05, np.pi, 2.1, -2.1][ci]
        R = rot_z(yaw) @ np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]]) @ rot_z(rng.normal(0, 0.01))
        t = np.array([np.cos(yaw) * 1.5, np.sin(yaw) * 0.8, 1.6]) + rng.normal(0, 0.05, 3)
        K = np.array([[1900.0 + ci, 0.0, 960.0 + rng.normal(0, 5)], [0.0, 1905.0 - ci, 540.0 + rng.normal(0, 5)], [0, 0, 1.0]])
        dist = rng.normal(0, 0.05, 5)
        cams[cname] = dict(data_path=f"/data/{cname}/{ci}.jpg", sensor2lidar_rotation=R, sensor2lidar_translation=t,
                           cam_intrinsic=K, cam_distortion=dist)
        cam_in.append(np.concatenate([R.ravel(), t, K.ravel(), dist]))
    info = dict(token="tok0", lidar_path="x.bin", sweeps=[], timestamp=1700000000000000, radars=radars, cams=cams)
    fake = types.SimpleNamespace(data_infos=[info], modality=dict(use_radar=True, use_camera=True), test_mode=True)
    d = ds.NewScenesDataset.get_data_info(fake, 0)
    out["cam_inputs"] = np.array(cam_in)                            # per camera: R(9), t(3), K(9), dist(5)
    out["cam_lidar2img"] = np.array(d["lidar2img"])
    out["cam_intrinsic"] = np.array(d["cam_intrinsic"])
    out["cam_lidar2cam"] = np.array(d["lidar2cam"])
    assert d["timestamp"] == 1700000000.0 and d["sample_idx"] == "tok0"

    from newscenes_devkit.eval.detection.config import config_factory
    cfg = config_factory("detection_newsc_config_final")
    n = 40
    boxes = np.concatenate([rng.uniform(-75, 75, (n, 1)), rng.uniform(-50, 50, (n, 1)), rng.uniform(-3, 1, (n, 1)),
                            rng.uniform(0.4, 9, (n, 3)), rng.uniform(-7, 7, (n, 1)), rng.normal(0, 4, (n, 2))], 1).astype(np.float32)
    det = dict(boxes_3d=_Boxes(torch.from_numpy(boxes)), scores_3d=torch.from_numpy(rng.uniform(0.05, 1, n).astype(np.float32)),
               labels_3d=torch.from_numpy(rng.integers(0, 4, n)))
    res = ds.output_to_newsc_box(det, ds.NewScenesDataset.CLASSES, cfg)
    out["o2n_boxes"], out["o2n_scores"], out["o2n_labels"] = boxes, det["scores_3d"].numpy(), det["labels_3d"].numpy()
    out["o2n_center"] = np.array([b.center for b in res])
    out["o2n_wlh"] = np.array([b.wlh for b in res])
    out["o2n_quat"] = np.array([b.orientation.elements for b in res])
    out["o2n_velocity"] = np.array([np.asarray(b.velocity, dtype=np.float64) for b in res])
    out["o2n_score"] = np.array([b.score for b in res])
    out["o2n_label"] = np.array([b.label for b in res])
    print("output_to_newsc_box:", n, "->", len(res))

    path = os.path.join(HERE, "data_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
