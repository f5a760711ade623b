"""Rank sharding of the frames (the "samples sharded across the GPUs" of the data-parallel step): the two samplers the
configs name (bevfusion.py:243-244) against index sequences produced by the reference classes
(tests/golden/make_golden_samplers.py), plus the properties a data-parallel job relies on."""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(os.path.dirname(__file__), "golden", "samplers_golden.json")) as f:
        return json.load(f)


class _Data:
    def __init__(self, flag):
        self.flag = np.asarray(flag, dtype=np.uint8)

    def __len__(self):
        return len(self.flag)


def test_group_sampler_index_sequences_are_the_reference_ones(gold):
    from projects.mmdet3d_plugin.datasets.samplers import SAMPLER, DistributedGroupSampler, build_sampler
    assert SAMPLER.get("DistributedGroupSampler") is DistributedGroupSampler
    for c in gold["group"]:
        data = _Data(gold["flags"][c["case"]])
        for rank in range(c["world"]):
            s = build_sampler(dict(type="DistributedGroupSampler"),
                              dict(dataset=data, samples_per_gpu=c["spg"], num_replicas=c["world"], rank=rank, seed=c["seed"]))
            s.set_epoch(c["epoch"])
            got = [int(i) for i in s]
            assert got == c["indices"][rank], (c["case"], c["world"], c["spg"], c["epoch"], rank)
            assert len(s) == len(got)


def test_test_time_sampler_gives_contiguous_blocks_like_the_reference(gold):
    from projects.mmdet3d_plugin.datasets.samplers import DistributedSampler
    for c in gold["dist"]:
        data = _Data(gold["flags"][c["case"]])
        for rank in range(c["world"]):
            s = DistributedSampler(data, num_replicas=c["world"], rank=rank, shuffle=False)
            assert [int(i) for i in s] == c["indices"][rank], (c["case"], c["world"], rank)
    with pytest.raises(AssertionError):
        iter(DistributedSampler(_Data([0] * 5), num_replicas=1, rank=0, shuffle=True)).__next__()


@pytest.mark.parametrize("world,spg", [(2, 1), (8, 1), (8, 2), (4, 3)])
def test_sharding_properties(world, spg):
    """Every rank gets the same number of frames, together they cover the dataset, a batch never mixes groups, every
    rank draws from the same permutation (seeded by epoch + seed), and epochs differ."""
    from projects.mmdet3d_plugin.datasets.samplers import DistributedGroupSampler
    rng = np.random.default_rng(world * 10 + spg)
    flag = rng.integers(0, 2, 211)
    data = _Data(flag)
    shards = {}
    for epoch in (0, 1):
        per_rank = []
        for r in range(world):
            s = DistributedGroupSampler(data, samples_per_gpu=spg, num_replicas=world, rank=r, seed=3)
            s.set_epoch(epoch)
            per_rank.append(list(s))
        shards[epoch] = per_rank
        assert len({len(p) for p in per_rank}) == 1 and len(per_rank[0]) % spg == 0
        everything = [i for p in per_rank for i in p]
        assert set(everything) == set(range(len(flag)))                       # covered (padding repeats a few)
        assert len(everything) - len(flag) < 2 * spg * world                  # ... at most one unit per group
        for p in per_rank:
            for b in range(0, len(p), spg):
                assert len({int(flag[i]) for i in p[b:b + spg]}) == 1         # a batch stays inside one group
    assert shards[0] != shards[1]


def test_samplers_take_rank_and_world_from_the_process_group_when_not_given():
    from projects.mmdet3d_plugin.datasets.samplers import DistributedGroupSampler, DistributedSampler
    s = DistributedGroupSampler(_Data([0] * 9), samples_per_gpu=2)
    assert (s.rank, s.num_replicas, len(s)) == (0, 1, 10)
    t = DistributedSampler(_Data([0] * 9), shuffle=False)
    assert list(t) == list(range(9))
