"""Temporal queue construction against the reference's queue dataset (tests/golden/make_golden_queue.py ran
``CustomNewScenesDataset.prepare_train_data`` / ``union2one``), and the ego-pose bookkeeping built on top of it."""
import json
import math
import os
import random

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(os.path.dirname(__file__), "golden", "queue_golden.json")) as f:
        return json.load(f)


def _frames(gold, picked):
    can = np.asarray(gold["can_bus"])
    return [dict(img=torch.full((2, 3), float(i)), img_metas=dict(scene_token=gold["scenes"][i], can_bus=can[i].copy(), idx=i))
            for i in picked]


def test_queue_indices_and_union2one_match_the_reference(gold):
    from projects.mmdet3d_plugin.datasets.temporal_queue import queue_indices, union2one
    for c in gold["cases"]:
        random.seed(c["seed"])
        picked = [max(0, i) for i in queue_indices(c["index"], 4)]
        assert picked == c["picked"], c["index"]
        out = union2one(_frames(gold, picked))
        assert out["img"][:, 0, 0].tolist() == c["img"] and out["img"].shape == (4, 2, 3)
        metas = out["img_metas"]
        assert [bool(metas[i]["prev_bev_exists"]) for i in range(4)] == c["prev_bev_exists"]
        for i in range(4):
            assert np.array_equal(metas[i]["can_bus"], np.asarray(c["can_bus"][i])), (c["index"], i)
    rng = random.Random(3)                                   # an own generator instead of the module-level one
    assert len(queue_indices(10, 4, rng)) == 4


def test_ego_deltas_recover_the_relative_poses_and_respect_scene_boundaries(gold):
    from projects.mmdet3d_plugin.bevfusion.detectors.bevf_triple_temporal import bev_warp_theta
    from projects.mmdet3d_plugin.datasets.temporal_queue import ego_deltas, union2one
    can = np.asarray(gold["can_bus"])
    picked = [3, 4, 5, 7]                                    # one scene
    yaw_last = can[7, -1]
    metas = union2one(_frames(gold, picked))["img_metas"]
    deltas, usable = ego_deltas(metas, yaw_last)
    assert usable == [True, True, True, True] and deltas[3] == (0.0, 0.0, 0.0)
    for t, i in enumerate(picked[:-1]):
        d = can[i, :2] - can[7, :2]
        a = -math.radians(yaw_last)
        want = (math.cos(a) * d[0] - math.sin(a) * d[1], math.sin(a) * d[0] + math.cos(a) * d[1], math.radians(can[i, -1] - yaw_last))
        assert np.allclose(deltas[t], want, atol=1e-9), t
    # a landmark seen from frame t lands where the last frame sees it
    i, t = picked[0], 0
    yaw_t = math.radians(can[i, -1])
    landmark = can[i, :2] + np.array([math.cos(yaw_t) * 5 - math.sin(yaw_t) * 2, math.sin(yaw_t) * 5 + math.cos(yaw_t) * 2])   # (5, 2) in frame t
    a = -math.radians(yaw_last)
    rel = landmark - can[7, :2]
    in_last = np.array([math.cos(a) * rel[0] - math.sin(a) * rel[1], math.sin(a) * rel[0] + math.cos(a) * rel[1]])
    dx, dy, dyaw = deltas[t]
    moved = np.array([math.cos(dyaw) * 5 - math.sin(dyaw) * 2 + dx, math.sin(dyaw) * 5 + math.cos(dyaw) * 2 + dy])
    assert np.allclose(moved, in_last, atol=1e-9)
    assert bev_warp_theta(deltas[t], [-60, -40, -3, 60, 40, 5]).shape == (2, 3)
    # scene boundary between queue positions 1 and 2: the two older frames are unusable
    picked = [9, 10, 11, 12]
    metas = union2one(_frames(gold, picked))["img_metas"]
    deltas, usable = ego_deltas(metas, can[12, -1])
    assert [metas[i]["prev_bev_exists"] for i in range(4)] == [False, True, False, True]
    assert usable == [False, False, True, True] and deltas[0] == deltas[1] == (0.0, 0.0, 0.0) and deltas[2] != (0.0, 0.0, 0.0)


def test_history_frames_of_another_scene_are_dropped_by_the_detector():
    from omnihd_amd.harness import FusionTrainStep
    from oracle.torch_shim import oracle_ops
    torch.set_num_threads(4)
    with oracle_ops():
        st = FusionTrainStep(res="tiny", batch=2, radar_dims=7, device="cpu", dtype="fp32", channels_last=False, sets=1,
                             task="triple", frames=3)
        m, b = st.raw_model, st.batches[0]
        full = m._history_bev(b["points"], b["lidar_points"], b["img"], b["img_metas"])
        b["img_metas"][1][0]["history_valid"] = False          # sample 1, oldest frame: another scene
        cut = m._history_bev(b["points"], b["lidar_points"], b["img"], b["img_metas"])
    assert torch.equal(cut[0], full[0]) and torch.equal(cut[1, 384:], full[1, 384:])
    assert float(cut[1, :384].abs().sum()) == 0.0 and float(full[1, :384].abs().sum()) > 0.0
