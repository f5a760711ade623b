"""Gaussian depth targets for the KL depth loss.

Mirrors ``generate_guassian_depth_target`` of the reference
(projects/mmdet3d_plugin/utils/gaussian.py:90-129; the function name keeps the reference's
spelling).  Same arithmetic, expressed without the 60-iteration Python loop over ``dist.cdf``:
the normal CDF is evaluated for all bin edges at once.
"""
import math

import torch
import torch.nn.functional as F

__all__ = ["generate_guassian_depth_target"]


def generate_guassian_depth_target(depth, stride, cam_depth_range, constant_std=None):
    """depth (B, N, tH, tW) sparse ground-truth depth (0 = no return) ->
    (depth_dist (B*N, H, W, D), min_depth (B*N, H, W)) with H = tH // stride."""
    depth = depth.flatten(0, 1)
    B, tH, tW = depth.shape
    H, W = tH // stride, tW // stride
    patches = F.unfold(depth.unsqueeze(1), stride, dilation=1, padding=0, stride=stride)   # B, k*k, H*W
    patches = patches.view(B, -1, H, W).permute(0, 2, 3, 1).contiguous()                   # B, H, W, k*k
    valid = patches != 0
    if constant_std is None:
        valid_f = valid.float()
        num = valid_f.sum(-1)
        num[num == 0] = 1e10
        mean = patches.sum(-1) / num
        var_sum = (((patches - mean.unsqueeze(-1)) ** 2) * valid_f).sum(-1)
        std = torch.sqrt(var_sum / num)
        std[num == 1] = 1
    else:
        std = torch.full((B, H, W), float(constant_std), dtype=torch.float32, device=depth.device)
    patches[~valid] = 1e10
    min_depth = patches.min(dim=-1)[0]
    min_depth[min_depth == 1e10] = 0
    # bin edges in raw depth (reference :119), CDF differences over consecutive edges (:121-127)
    edges = torch.arange(cam_depth_range[0] - cam_depth_range[2] / 2, cam_depth_range[1], cam_depth_range[2],
                         device=depth.device)
    loc = (min_depth / cam_depth_range[2]).unsqueeze(-1)
    scale = (std / cam_depth_range[2]).unsqueeze(-1)
    cdf = 0.5 * (1 + torch.erf((edges.view(1, 1, 1, -1) - loc) * scale.reciprocal() / math.sqrt(2)))
    return cdf[..., 1:] - cdf[..., :-1], min_depth
