"""How frames are sharded over the ranks at TEST time — mirror of
projects/mmdet3d_plugin/datasets/samplers/distributed_sampler.py:8-41.

No shuffling (the reference asserts on it); the index list 0..len-1 is repeated up to ``total_size`` =
ceil(len / world) * world and rank r takes the CONTIGUOUS block [r * per, (r + 1) * per) — not torch's strided
``indices[rank::world]``: contiguous blocks keep the frames of a sequence on one rank, which the temporal models of
the repo need, and make the gathered results come back in dataset order."""
import math

from torch.utils.data import DistributedSampler as _TorchDistributedSampler

from ._dist import get_dist_info
from .sampler import SAMPLER


@SAMPLER.register_module()
class DistributedSampler(_TorchDistributedSampler):
    def __init__(self, dataset=None, num_replicas=None, rank=None, shuffle=True, seed=0):
        if num_replicas is None or rank is None:            # torch asks the default process group; stay usable without one
            r, w = get_dist_info()
            num_replicas = w if num_replicas is None else num_replicas
            rank = r if rank is None else rank
        super().__init__(dataset, num_replicas=num_replicas, rank=rank, shuffle=shuffle)
        self.seed = 0 if seed is None else seed

    def __iter__(self):
        assert not self.shuffle, "the reference's test-time sampler does not shuffle (distributed_sampler.py:24-25)"
        n = len(self.dataset)
        order = (list(range(n)) * math.ceil(self.total_size / n))[:self.total_size]
        per = self.total_size // self.num_replicas
        mine = order[self.rank * per:(self.rank + 1) * per]
        assert len(mine) == self.num_samples
        return iter(mine)
