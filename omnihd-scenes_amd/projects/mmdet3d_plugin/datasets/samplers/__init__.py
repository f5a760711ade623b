from .distributed_sampler import DistributedSampler  # noqa: F401
from .group_sampler import DistributedGroupSampler  # noqa: F401
from .sampler import SAMPLER, build_sampler  # noqa: F401
