#!/usr/bin/env python3
"""Golden vectors for ``RadarPoints`` (projects/mmdet3d_plugin/core/points/radar_points.py:5-98): flip / scale /
rotate / in_range_bev run on the reference class over an inert ``BasePoints`` stand-in (stores the tensor; the four
methods read nothing else).  Usage: python tests/golden/make_golden_points.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_data as G  # noqa: E402


class BasePoints:
    def __init__(self, tensor, points_dim=3, attribute_dims=None):
        self.tensor, self.points_dim, self.attribute_dims = torch.as_tensor(tensor, dtype=torch.float32).clone(), points_dim, attribute_dims


def main():
    G._mod("mmdet3d")
    G._mod("mmdet3d.core")
    G._mod("mmdet3d.core.points")
    G._mod("mmdet3d.core.points.base_points", BasePoints=BasePoints)
    rp = G.load_file("ref_radar_points", "projects/mmdet3d_plugin/core/points/radar_points.py")
    rng = np.random.default_rng(8)
    pts = rng.normal(0, 10, (40, 8)).astype(np.float32)
    out = {"pts": pts}
    for d in ("horizontal", "vertical"):
        p = rp.RadarPoints(pts, points_dim=8)
        p.flip(d)
        out[f"flip_{d}"] = p.tensor.numpy()
    p = rp.RadarPoints(pts, points_dim=8)
    p.scale(1.25)
    out["scale"] = p.tensor.numpy()
    for tag, (rot, axis) in {"z": (0.3, None), "y": (-0.7, 1), "x": (1.1, 0), "m1": (0.5, -1)}.items():
        p = rp.RadarPoints(pts, points_dim=8)
        out[f"rot_{tag}_T"] = np.asarray(p.rotate(rot, axis))
        out[f"rot_{tag}"] = p.tensor.numpy()
    m = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
    p = rp.RadarPoints(pts, points_dim=8)
    out["rot_mat_T"] = np.asarray(p.rotate(torch.from_numpy(m)))
    out["rot_mat_in"], out["rot_mat"] = m, p.tensor.numpy()
    out["bev_range"] = np.array([-5.0, -8.0, 7.0, 9.0])
    out["in_bev"] = rp.RadarPoints(pts, points_dim=8).in_range_bev(out["bev_range"].tolist()).numpy()
    path = os.path.join(HERE, "points_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
