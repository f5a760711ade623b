"""Dataset-side contracts either side of the detector — mirror of
the reference's ``NewScenesDataset`` (projects/mmdet3d_plugin/datasets/newscenes_dataset.py):
``get_data_info`` :164-234 (the lidar2img / cam_intrinsic composition that becomes ``img_metas``),
``get_ann_info`` :236-283, ``_format_bbox`` :285-331, ``_evaluate_single`` :333-388,
``format_results`` / ``evaluate`` :390-478 and ``output_to_newsc_box`` :537-583.

The info records are taken in memory (``data_infos`` list or a pickle path); nothing here reads the
1.3 TB dataset.  Evaluation runs against ground truth built from the same info records
(``gt_eval_boxes``) with the devkit restatement in ``newscenes_devkit.eval``.
"""
import json
import os
import pickle
import tempfile

import numpy as np
import torch

from newscenes_devkit.eval.common.data_classes import EvalBoxes
from newscenes_devkit.eval.common.loaders import yaw_to_wxyz
from newscenes_devkit.eval.detection.config import config_factory
from newscenes_devkit.eval.detection.data_classes import DetectionBox
from newscenes_devkit.eval.detection.evaluate import NewScenesEval
from omnihd_amd.mm.boxes import LiDARInstance3DBoxes


def camera_matrices(cam_info):
    """One camera's info record -> (lidar2img 4x4, intrinsic 4x4 'viewpad', lidar2cam 4x4), float64."""
    lidar2cam_r = np.linalg.inv(cam_info["sensor2lidar_rotation"])
    lidar2cam_t = cam_info["sensor2lidar_translation"] @ lidar2cam_r.T
    rt = np.eye(4)
    rt[:3, :3] = lidar2cam_r.T
    rt[3, :3] = -lidar2cam_t
    intrinsic = np.array(cam_info["cam_intrinsic"])
    viewpad = np.eye(4)
    viewpad[:intrinsic.shape[0], :intrinsic.shape[1]] = intrinsic
    return viewpad @ rt.T, viewpad, rt.T


def output_to_newsc_box(detection, classes, eval_configs):
    """Detector output of one sample -> list of box records in the benchmark's convention: gravity
    centre, (x_size, y_size, z_size) as wlh, heading ``-yaw - pi/2`` as a z-axis quaternion, velocity
    from columns 7:9; boxes beyond the class's |x|,|y| evaluation range are dropped."""
    box3d = detection["boxes_3d"]
    scores = detection["scores_3d"].numpy()
    labels = detection["labels_3d"].numpy()
    centre = box3d.gravity_center.numpy()
    dims = box3d.dims.numpy()
    yaw = -box3d.yaw.numpy() - np.pi / 2
    out = []
    for i in range(len(box3d)):
        rng = eval_configs.class_range[classes[labels[i]]]
        if abs(centre[i][0]) > rng[0] or abs(centre[i][1]) > rng[1]:
            continue
        vel = (float(box3d.tensor[i, 7]), float(box3d.tensor[i, 8]), 0.0)
        out.append(dict(center=centre[i], wlh=dims[i], orientation=yaw_to_wxyz(yaw[i]), label=int(labels[i]),
                        score=float(scores[i]), velocity=vel))
    return out


class NewScenesDataset:
    NameMapping = {"suv": "car", "van": "car", "truck": "large_vehicle", "rider": "rider", "pedestrian": "pedestrian",
                   "car": "car", "tricyclist": "car", "light_truck": "large_vehicle", "bus": "large_vehicle",
                   "engineering_vehicle": "large_vehicle", "handcart": "car", "trailer": "large_vehicle"}
    ErrNameMapping = {"trans_err": "mATE", "scale_err": "mASE", "orient_err": "mAOE", "vel_err": "mAVE"}
    CLASSES = ("car", "pedestrian", "rider", "large_vehicle")

    def __init__(self, ann_file=None, pipeline=None, data_root=None, classes=None, load_interval=1, with_velocity=True,
                 modality=None, box_type_3d="LiDAR", filter_empty_gt=True, test_mode=False,
                 eval_version="detection_newsc_config_final", use_valid_flag=False, data_infos=None, metadata=None):
        self.load_interval, self.use_valid_flag, self.with_velocity = load_interval, use_valid_flag, with_velocity
        self.test_mode, self.data_root, self.pipeline = test_mode, data_root, pipeline
        if classes is not None:
            self.CLASSES = tuple(classes)
        self.cat2id = {name: i for i, name in enumerate(self.CLASSES)}
        if data_infos is None:
            data_infos, metadata = self.load_annotations(ann_file)
        else:
            data_infos = list(sorted(data_infos, key=lambda e: e["timestamp"]))[::load_interval]
        self.data_infos, self.metadata = data_infos, metadata or {"version": "v1.0-trainval"}
        self.version = self.metadata["version"]
        self.eval_version = eval_version
        self.eval_detection_configs = config_factory(eval_version)
        self.modality = modality or dict(use_camera=True, use_lidar=False, use_radar=True, use_map=False, use_external=False)

    def __len__(self):
        return len(self.data_infos)

    def load_annotations(self, ann_file):
        with open(ann_file, "rb") as f:
            data = pickle.load(f)
        infos = list(sorted(data["infos"], key=lambda e: e["timestamp"]))[::self.load_interval]
        return infos, data["metadata"]

    def get_cat_ids(self, idx):
        info = self.data_infos[idx]
        mask = info["valid_flag"] if self.use_valid_flag else np.full_like(info["valid_flag"], True)
        return [self.cat2id[n] for n in set(info["gt_names"][mask]) if n in self.CLASSES]

    def get_data_info(self, index):
        info = self.data_infos[index]
        d = dict(sample_idx=info["token"], pts_filename=info["lidar_path"], sweeps=info["sweeps"],
                 timestamp=int(info["timestamp"]) / 1e6)
        if self.modality["use_radar"]:
            d["radars"] = info["radars"]
        if self.modality["use_camera"]:
            mats = [camera_matrices(c) for c in info["cams"].values()]
            d.update(img_filename=[c["data_path"] for c in info["cams"].values()], lidar2img=[m[0] for m in mats],
                     cam_intrinsic=[m[1] for m in mats], lidar2cam=[m[2] for m in mats],
                     cam_distortion=[np.array(c["cam_distortion"]) for c in info["cams"].values()])
        if not self.test_mode:
            d["ann_info"] = self.get_ann_info(index)
        return d

    def get_ann_info(self, index):
        info = self.data_infos[index]
        mask = info["valid_flag"] if self.use_valid_flag else np.full_like(info["valid_flag"], True)
        boxes, names = info["gt_boxes"][mask], info["gt_names"][mask]
        labels = np.array([self.CLASSES.index(n) if n in self.CLASSES else -1 for n in names])
        if self.with_velocity:
            vel = info["gt_velocity"][mask]
            vel[np.isnan(vel[:, 0])] = [0.0, 0.0]
            boxes = np.concatenate([boxes, vel], axis=-1)
        boxes = LiDARInstance3DBoxes(torch.as_tensor(boxes, dtype=torch.float32), box_dim=boxes.shape[-1],
                                     origin=(0.5, 0.5, 0.5))
        return dict(gt_bboxes_3d=boxes, gt_labels_3d=labels, gt_names=names)

    # ---- results -> benchmark records -------------------------------------------------------
    def _format_bbox(self, results, jsonfile_prefix=None):
        annos = {}
        for sample_id, det in enumerate(results):
            token = self.data_infos[sample_id]["token"]
            annos[token] = [dict(sample_token=token, translation=b["center"].tolist(), size=b["wlh"].tolist(),
                                 rotation=list(b["orientation"]), velocity=list(b["velocity"][:2]),
                                 detection_name=self.CLASSES[b["label"]], detection_score=b["score"])
                            for b in output_to_newsc_box(det, self.CLASSES, self.eval_detection_configs)]
        submission = {"meta": self.modality, "results": annos}
        if jsonfile_prefix is None:
            return submission
        os.makedirs(jsonfile_prefix, exist_ok=True)
        path = os.path.join(jsonfile_prefix, "results_newsc.json")
        with open(path, "w") as f:
            json.dump(submission, f)
        return path

    def format_results(self, results, jsonfile_prefix=None):
        assert isinstance(results, list), "results must be a list"
        assert len(results) == len(self), "The length of results is not equal to the dataset len: {} != {}".format(
            len(results), len(self))
        tmp_dir = None
        if jsonfile_prefix is None:
            tmp_dir = tempfile.TemporaryDirectory()
            jsonfile_prefix = os.path.join(tmp_dir.name, "results")
        if not ("pts_bbox" in results[0] or "img_bbox" in results[0]):
            return self._format_bbox(results, jsonfile_prefix), tmp_dir
        files = {name: self._format_bbox([out[name] for out in results], os.path.join(jsonfile_prefix, name))
                 for name in results[0]}
        return files, tmp_dir

    def gt_eval_boxes(self):
        """Ground truth of every sample as evaluation boxes (what reference ``load_gt`` builds from the
        database: gravity centre, wlh, heading quaternion, visibility from ``valid_flag``), class names
        mapped through ``NameMapping``; categories outside it are skipped."""
        gt = EvalBoxes()
        for info in self.data_infos:
            boxes = []
            vel = info.get("gt_velocity")
            for i, name in enumerate(info["gt_names"]):
                det_name = self.NameMapping.get(name, name if name in self.CLASSES else None)
                if det_name is None:
                    continue
                b = np.asarray(info["gt_boxes"][i], dtype=float)
                v = (0.0, 0.0) if vel is None or np.isnan(vel[i][0]) else (float(vel[i][0]), float(vel[i][1]))
                centre = tuple(b[:3].tolist())
                boxes.append(DetectionBox(sample_token=info["token"], translation=centre, size=tuple(b[3:6].tolist()),
                                          rotation=yaw_to_wxyz(-b[6] - np.pi / 2), velocity=v, ego_translation=centre,
                                          detection_name=det_name, detection_score=-1.0,
                                          visibility=int(bool(info["valid_flag"][i]))))
            gt.add_boxes(info["token"], boxes)
        return gt

    def _evaluate_single(self, result_path, logger=None, metric="bbox", result_name="pts_bbox"):
        out_dir = os.path.dirname(result_path) if isinstance(result_path, str) else None
        summary = NewScenesEval(self.gt_eval_boxes(), self.eval_detection_configs, result_path, output_dir=out_dir,
                                verbose=False).main(render_curves=False)
        prefix = f"{result_name}_NewScenes"
        detail = {}
        for name in self.CLASSES:
            for k, v in summary["label_aps"][name].items():
                detail["{}/{}_AP_dist_{}".format(prefix, name, k)] = float("{:.4f}".format(v))
            for k, v in summary["label_tp_errors"][name].items():
                detail["{}/{}_{}".format(prefix, name, k)] = float("{:.4f}".format(v))
            for k, v in summary["tp_errors"].items():
                detail["{}/{}".format(prefix, self.ErrNameMapping[k])] = float("{:.4f}".format(v))
        detail["{}/NOS".format(prefix)] = summary["NOS"]
        detail["{}/mAP".format(prefix)] = summary["mean_ap"]
        return detail

    def evaluate(self, results, metric="bbox", logger=None, jsonfile_prefix=None, result_names=("pts_bbox",), show=False,
                 out_dir=None, pipeline=None):
        files, tmp_dir = self.format_results(results, jsonfile_prefix)
        if isinstance(files, dict):
            out = {}
            for name in result_names:
                out.update(self._evaluate_single(files[name], result_name=name))
        else:
            out = self._evaluate_single(files)
        if tmp_dir is not None:
            tmp_dir.cleanup()
        return out
