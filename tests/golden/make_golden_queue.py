#!/usr/bin/env python3
"""Golden vectors for the temporal queue construction: ``CustomNewScenesDataset.prepare_train_data`` / ``union2one``
(projects/mmdet3d_plugin/datasets/custom_newscenes_dataset.py:28-85) run on an instance made without its constructor
(``object.__new__``) with identity stand-ins for ``get_data_info`` / ``pre_pipeline`` / ``pipeline``; absent packages
get inert stand-in modules.  Usage: python tests/golden/make_golden_queue.py"""
import copy
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_data as G  # noqa: E402


class DC:
    def __init__(self, data, cpu_only=False, stack=False):
        self.data, self._data = data, data


def main():
    G.install_stubs()
    G._mod("pyquaternion")
    G._mod("newscenes_devkit")
    G._mod("newscenes_devkit.data_classes", Box=object)
    G._mod("newscenes_devkit.eval")
    G._mod("newscenes_devkit.eval.common")
    G._mod("newscenes_devkit.eval.common.utils", quaternion_yaw=None, Quaternion=None)
    G._mod("mmcv.parallel", DataContainer=DC)
    sys.modules["projects.mmdet3d_plugin.datasets"] = G._mod("projects.mmdet3d_plugin.datasets", NewScenesDataset=object)
    ref = G.load_file("ref_custom_ds", "projects/mmdet3d_plugin/datasets/custom_newscenes_dataset.py")
    ds = object.__new__(ref.CustomNewScenesDataset)
    rng = np.random.default_rng(12)
    n = 30
    scenes = ["a"] * 11 + ["b"] * 9 + ["c"] * 10
    can = np.zeros((n, 18))
    can[:, :3] = np.cumsum(rng.normal(0, 1.5, (n, 3)), 0)
    can[:, -1] = (np.cumsum(rng.normal(0, 4.0, n)) + 350) % 360
    can[:, -2] = can[:, -1] / 180 * np.pi
    ds.queue_length, ds.filter_empty_gt = 4, False
    ds.get_data_info = lambda i: dict(idx=i)
    ds.pre_pipeline = lambda d: None
    ds.pipeline = lambda d: dict(img=DC(torch.full((2, 3), float(d["idx"]))),
                                 img_metas=DC(dict(scene_token=scenes[d["idx"]], can_bus=can[d["idx"]].copy(), idx=d["idx"])))
    out = dict(scenes=scenes, can_bus=can.tolist(), cases=[])
    for index, seed in [(0, 1), (2, 2), (7, 3), (11, 4), (12, 5), (13, 6), (21, 7), (29, 8)]:
        random.seed(seed)
        res = ds.prepare_train_data(index)
        metas = res["img_metas"].data
        out["cases"].append(dict(index=index, seed=seed, picked=[int(metas[i]["idx"]) for i in range(len(metas))],
                                 img=res["img"].data[:, 0, 0].tolist(),
                                 can_bus=[metas[i]["can_bus"].tolist() for i in range(len(metas))],
                                 prev_bev_exists=[bool(metas[i]["prev_bev_exists"]) for i in range(len(metas))]))
    path = os.path.join(HERE, "queue_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, len(out["cases"]), "cases;", [c["picked"] for c in out["cases"]])


if __name__ == "__main__":
    main()
