"""How frames are sharded over the ranks at TRAINING time — mirror of
projects/mmdet3d_plugin/datasets/samplers/group_sampler.py:12-109 (``DistributedGroupSampler``): this is the
"samples sharded across the GPUs" of the data-parallel step.

Per epoch, with ONE generator seeded ``epoch + seed`` on every rank (so all ranks draw the same permutation):
every aspect-ratio group (``dataset.flag``) is shuffled, padded by repetition to a multiple of
``samples_per_gpu * world``, the groups are concatenated, the resulting list is shuffled again in units of
``samples_per_gpu`` consecutive entries (a batch never mixes groups), and rank r takes the contiguous slice
[r * num_samples, (r + 1) * num_samples).  The permutations come from ``torch.randperm`` on the CPU generator, as in
the reference, so that the index sequences are the reference's for the same torch build."""
import math

import numpy as np
import torch
from torch.utils.data import Sampler

from ._dist import get_dist_info
from .sampler import SAMPLER


@SAMPLER.register_module()
class DistributedGroupSampler(Sampler):
    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None, seed=0):
        r, w = get_dist_info()
        self.dataset, self.samples_per_gpu = dataset, samples_per_gpu
        self.num_replicas = w if num_replicas is None else num_replicas
        self.rank = r if rank is None else rank
        self.epoch, self.seed = 0, (0 if seed is None else seed)
        assert hasattr(dataset, "flag")
        self.flag = dataset.flag
        self.group_sizes = np.bincount(self.flag)
        unit = self.samples_per_gpu * self.num_replicas
        self._padded = [int(math.ceil(int(s) / unit)) * unit for s in self.group_sizes]
        self.num_samples = sum(p // self.num_replicas for p in self._padded)
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch + self.seed)
        order = []
        for group, (size, padded) in enumerate(zip(self.group_sizes, self._padded)):
            size = int(size)
            if size == 0:
                continue
            members = np.where(self.flag == group)[0]
            shuffled = members[torch.randperm(size, generator=g).numpy()].tolist()
            extra = padded - size
            order += shuffled + shuffled * (extra // size) + shuffled[:extra % size]
        assert len(order) == self.total_size
        spg = self.samples_per_gpu
        batches = torch.randperm(len(order) // spg, generator=g).tolist()
        order = [order[b * spg + j] for b in batches for j in range(spg)]
        mine = order[self.num_samples * self.rank:self.num_samples * (self.rank + 1)]
        assert len(mine) == self.num_samples
        return iter(mine)

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
