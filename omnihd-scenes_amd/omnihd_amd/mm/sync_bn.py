"""naiveSyncBN1d/2d/3d — per-rank mean / mean-of-squares averaged over ranks.

Reference: projects/mmdet3d_plugin/ops/norm.py:9-82 (3-D variant, in the repo) and the same
algorithm for 1-D/2-D in upstream mmdet3d (not vendored).  Semantics kept: statistics are the
MEAN OF PER-RANK MEANS (not count-weighted, norm.py:65-72); plain BatchNorm when there is no
process group, one rank, or eval mode (norm.py:58).  The exchange uses a differentiable
all-reduce(SUM) — forward value and backward gradient are the same as the reference's
all_gather+sum forward / all_reduce backward (norm.py:12-24), one collective instead of
world_size buffers; over RCCL the 2*C-float message is latency-bound either way.
"""
import torch
from torch import distributed as dist
from torch import nn
from torch.autograd.function import Function


class AllReduceSum(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.clone()
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
        return x

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g


class _NaiveSyncBN(nn.modules.batchnorm._BatchNorm):
    _reduce_dims = None   # dims averaged locally
    _omnihd_sync = True   # bricks.bn_act: statistics of this layer are averaged over the ranks

    def _check_input_dim(self, input):
        pass

    def forward(self, input):
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1 or not self.training:
            return super().forward(input)
        assert input.shape[0] > 0, "SyncBN does not support empty inputs"
        from .. import ops
        if ops.bn_train_supported(input) and self.momentum is not None and self.track_running_stats:
            # fused kernels: one pass for the statistics, the exchange, one pass to normalise (csrc/batch_norm.hip)
            if self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
            return ops.bn_train_act(input, self.weight, self.bias, self.running_mean, self.running_var, self.momentum,
                                    self.eps, False, dist.group.WORLD)
        x = input.float()
        C = x.shape[1]
        dims = [d for d in range(x.dim()) if d != 1]
        mean = torch.mean(x, dim=dims)
        meansqr = torch.mean(x * x, dim=dims)
        vec = AllReduceSum.apply(torch.cat([mean, meansqr], dim=0)) * (1.0 / dist.get_world_size())
        mean, meansqr = torch.split(vec, C)
        var = meansqr - mean * mean
        with torch.no_grad():
            self.running_mean += self.momentum * (mean.detach() - self.running_mean)
            self.running_var += self.momentum * (var.detach() - self.running_var)
        invstd = torch.rsqrt(var + self.eps)
        scale = self.weight * invstd
        bias = self.bias - mean * scale
        shape = [1, -1] + [1] * (x.dim() - 2)
        return (x * scale.reshape(shape) + bias.reshape(shape)).to(input.dtype)


class NaiveSyncBatchNorm1d(_NaiveSyncBN):
    """Inputs (N, C) or (N, C, L)."""


class NaiveSyncBatchNorm2d(_NaiveSyncBN):
    """Inputs (N, C, H, W)."""


class NaiveSyncBatchNorm3d(_NaiveSyncBN):
    """Inputs (N, C, D, H, W) — reference ops/norm.py:28-82."""
