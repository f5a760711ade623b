#!/usr/bin/env python3
"""Golden index sequences of the reference's rank-sharding samplers
(projects/mmdet3d_plugin/datasets/samplers/{group_sampler,distributed_sampler}.py), produced by running the reference
classes with stand-ins for mmcv's registry / ``get_dist_info`` and IPython (none carries arithmetic).  The sequences
depend on ``torch.randperm`` of the CPU generator, i.e. on the torch build — the same build here and on the GPU box.
Usage: python tests/golden/make_golden_samplers.py"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_data as G  # noqa: E402


class _Reg:
    def __init__(self, name):
        self.name = name

    def register_module(self, *a, **k):
        return (lambda c: c) if not (len(a) == 1 and callable(a[0])) else a[0]


def main():
    G._mod("mmcv")
    G._mod("mmcv.utils")
    G._mod("mmcv.utils.registry", Registry=_Reg, build_from_cfg=None)
    G._mod("mmcv.runner", get_dist_info=lambda: (0, 1))
    G._mod("IPython", embed=None)
    pkg = types.ModuleType("ref_samplers")
    pkg.__path__ = [os.path.join(G.REF, "projects/mmdet3d_plugin/datasets/samplers")]
    sys.modules["ref_samplers"] = pkg
    G.load_file("ref_samplers.sampler", "projects/mmdet3d_plugin/datasets/samplers/sampler.py")
    gs = G.load_file("ref_samplers.group_sampler", "projects/mmdet3d_plugin/datasets/samplers/group_sampler.py")
    ds = G.load_file("ref_samplers.distributed_sampler", "projects/mmdet3d_plugin/datasets/samplers/distributed_sampler.py")

    class Data:
        def __init__(self, flag):
            self.flag = np.asarray(flag, dtype=np.uint8)

        def __len__(self):
            return len(self.flag)

    rng = np.random.default_rng(4)
    cases = {"one_group_37": np.zeros(37, np.uint8).tolist(), "two_groups_50": rng.integers(0, 2, 50).tolist(),
             "gap_group_23": (rng.integers(0, 2, 23) * 2).tolist(), "tiny_3": [0, 0, 0]}
    out = {"flags": cases, "group": [], "dist": []}
    for name, flag in cases.items():
        for world, spg, seed in [(1, 1, 0), (2, 1, 0), (8, 1, 0), (8, 2, 7), (4, 3, 1)]:
            for epoch in (0, 5):
                per_rank = []
                for rank in range(world):
                    s = gs.DistributedGroupSampler(Data(flag), samples_per_gpu=spg, num_replicas=world, rank=rank, seed=seed)
                    s.set_epoch(epoch)
                    per_rank.append([int(i) for i in s])
                    assert len(per_rank[-1]) == len(s)
                out["group"].append(dict(case=name, world=world, spg=spg, seed=seed, epoch=epoch, indices=per_rank))
        for world in (1, 2, 8):
            per_rank = [[int(i) for i in ds.DistributedSampler(Data(flag), num_replicas=world, rank=r, shuffle=False)]
                        for r in range(world)]
            out["dist"].append(dict(case=name, world=world, indices=per_rank))
    path = os.path.join(HERE, "samplers_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out["group"]), "+", len(out["dist"]), "cases")


if __name__ == "__main__":
    main()
