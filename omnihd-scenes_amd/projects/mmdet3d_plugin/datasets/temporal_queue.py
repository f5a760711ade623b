"""Queue construction for the temporal detector — the two pieces of the reference's queue dataset
(projects/mmdet3d_plugin/datasets/custom_newscenes_dataset.py, ``CustomNewScenesDataset``) that decide WHICH frames
form a training queue and HOW their ego poses are related, restated as functions over plain data:

* ``queue_indices`` (:38-42): of the ``queue_length`` frames before ``index`` one is dropped at random, the rest are
  kept in order, ``index`` itself closes the queue; indices are clamped at 0 by the caller (:44);
* ``union2one`` (:58-85): metas keyed by queue position; ``can_bus[:3]`` (position) and ``can_bus[-1]`` (yaw in degrees)
  become DELTAS to the previous frame of the queue, zero where a new scene starts, and ``prev_bev_exists`` says whether
  a frame continues the scene of its predecessor.

``ego_deltas`` is ours: what ``BEVFusionTripleTemporal`` needs from those metas — the pose of every history frame in
the LAST frame's ego/LiDAR frame, accumulated from the per-step deltas (positions are global, headings are the
``can_bus`` yaw), and which history frames are usable (no scene boundary between them and the last frame)."""
import copy
import math
import random

import numpy as np
import torch

__all__ = ["queue_indices", "union2one", "ego_deltas"]


def queue_indices(index, queue_length, rng=random):
    """Positions of the frames of one training queue (``rng``: the ``random`` module or a ``random.Random``)."""
    before = list(range(index - queue_length, index))
    rng.shuffle(before)
    return sorted(before[1:]) + [index]


def union2one(queue):
    """``queue``: list of dicts with ``img`` (tensor) and ``img_metas`` (dict with ``scene_token`` and ``can_bus``), oldest
    first.  Returns the last entry with ``img`` stacked over the queue and ``img_metas`` = {position: meta}; the metas'
    ``can_bus`` arrays are modified in place exactly as the reference does."""
    metas = {}
    scene = last_pos = last_angle = None
    for i, frame in enumerate(queue):
        m = metas[i] = frame["img_metas"]
        pos, angle = copy.deepcopy(m["can_bus"][:3]), copy.deepcopy(m["can_bus"][-1])
        if m["scene_token"] != scene:
            m["prev_bev_exists"] = False
            scene = m["scene_token"]
            m["can_bus"][:3] = 0
            m["can_bus"][-1] = 0
        else:
            m["prev_bev_exists"] = True
            m["can_bus"][:3] -= last_pos
            m["can_bus"][-1] -= last_angle
        last_pos, last_angle = pos, angle
    out = queue[-1]
    out["img"] = torch.stack([f["img"] for f in queue])
    out["img_metas"] = metas
    return out


def ego_deltas(metas, yaw_of_last_deg):
    """``metas`` = the {position: meta} map of ``union2one``; ``yaw_of_last_deg`` = absolute heading of the last frame
    (its ``can_bus[-1]`` BEFORE ``union2one`` turned it into a delta).  -> (deltas, usable): ``deltas[t]`` = (dx, dy,
    dyaw) of frame t's ego pose expressed in the last frame (what ``bev_warp_theta`` takes), ``usable[t]`` False for
    frames separated from the last one by a scene boundary."""
    T = len(metas)
    deltas, usable = [(0.0, 0.0, 0.0)] * T, [True] * T
    # walk back from the last frame, undoing one step at a time; global offsets are rotated into the last frame at the end
    off = np.zeros(2)
    dyaw_deg = 0.0
    ok = True
    yaw_last = math.radians(yaw_of_last_deg)
    c, s = math.cos(-yaw_last), math.sin(-yaw_last)
    for t in range(T - 2, -1, -1):
        step = metas[t + 1]
        ok = ok and bool(step.get("prev_bev_exists", True))
        off = off - np.asarray(step["can_bus"][:2], dtype=np.float64)          # position of frame t minus position of frame T-1
        dyaw_deg -= float(step["can_bus"][-1])
        usable[t] = ok
        deltas[t] = (float(c * off[0] - s * off[1]), float(s * off[0] + c * off[1]), math.radians(dyaw_deg)) if ok else (0.0, 0.0, 0.0)
    return deltas, usable
