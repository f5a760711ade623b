"""``SAMPLER`` registry of the plugin (reference: datasets/samplers/sampler.py:1-7); the configs name the samplers by
``shuffler_sampler=dict(type='DistributedGroupSampler')`` / ``nonshuffler_sampler=dict(type='DistributedSampler')``
(bevfusion.py:243-244)."""
from omnihd_amd.mm.registry import Registry, build_from_cfg

SAMPLER = Registry("sampler")


def build_sampler(cfg, default_args):
    return build_from_cfg(cfg, SAMPLER, default_args)
