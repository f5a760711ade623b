"""Layer builders with mmcv's call signatures: build_norm_layer, build_conv_layer, ConvModule."""
import os
from .._env import env as _env

import torch
import torch.nn.functional as F
from torch import nn

from .registry import NORM_LAYERS


def _register_default_norms():
    from .sync_bn import NaiveSyncBatchNorm1d, NaiveSyncBatchNorm2d, NaiveSyncBatchNorm3d
    for name, cls in [("BN", nn.BatchNorm2d), ("BN1d", nn.BatchNorm1d), ("BN2d", nn.BatchNorm2d),
                      ("BN3d", nn.BatchNorm3d), ("SyncBN", nn.SyncBatchNorm), ("GN", nn.GroupNorm),
                      ("naiveSyncBN1d", NaiveSyncBatchNorm1d), ("naiveSyncBN2d", NaiveSyncBatchNorm2d),
                      ("naiveSyncBN3d", NaiveSyncBatchNorm3d)]:
        if name not in NORM_LAYERS:
            NORM_LAYERS.register_module(name=name, module=cls)


_ABBR = {"BN": "bn", "BN1d": "bn", "BN2d": "bn", "BN3d": "bn", "SyncBN": "bn", "GN": "gn",
         "naiveSyncBN1d": "bn", "naiveSyncBN2d": "bn", "naiveSyncBN3d": "bn"}


def build_norm_layer(cfg, num_features, postfix=""):
    """-> (name, layer); cfg = dict(type=..., requires_grad=True, eps=..., momentum=...)."""
    _register_default_norms()
    cfg = dict(cfg)
    t = cfg.pop("type")
    cls = NORM_LAYERS.get(t)
    if cls is None:
        raise KeyError(f"unknown norm layer type {t}")
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)
    if t == "GN":
        layer = cls(num_channels=num_features, **cfg)
    else:
        layer = cls(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return _ABBR.get(t, "norm") + str(postfix), layer


def build_conv_layer(cfg, *args, **kwargs):
    """cfg None / dict(type='Conv2d') -> nn.Conv2d; dict(type='DCN', ...) -> DeformConv2dPack.
    Extra keys of cfg are forwarded to the layer, as mmcv does."""
    if cfg is None:
        return nn.Conv2d(*args, **kwargs)
    cfg = dict(cfg)
    t = cfg.pop("type")
    if t in ("Conv2d", "Conv"):
        return nn.Conv2d(*args, **kwargs, **cfg)
    if t == "DCN":
        from .dcn import DeformConv2dPack
        return DeformConv2dPack(*args, **kwargs, **cfg)
    raise KeyError(f"unknown conv layer type {t}")


def build_activation(cfg):
    cfg = dict(cfg)
    t = cfg.pop("type")
    if t == "ReLU":
        return nn.ReLU(**cfg)
    if t == "Sigmoid":
        return nn.Sigmoid()
    raise KeyError(f"unknown activation {t}")


def frozen_bn_constants(bn):
    """(scale, shift) fp32 of a BatchNorm that is in eval mode with constant affine parameters:
    y = x * scale + shift.  Cached on the module; rebuilt when any of its tensors is written to."""
    tensors = (bn.running_mean, bn.running_var, bn.weight, bn.bias)
    key = tuple((t.data_ptr(), t._version) for t in tensors)
    cached = getattr(bn, "_omnihd_affine", None)
    if cached is None or cached[0] != key:
        with torch.no_grad():
            scale = (bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)).contiguous()
            shift = (bn.bias.float() - bn.running_mean.float() * scale).contiguous()
        cached = (key, scale, shift)
        bn._omnihd_affine = cached
    return cached[1], cached[2]


def _flush_batches_tracked(bn, *_):
    n = getattr(bn, "_omnihd_pending_batches", 0)
    if n and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(n)
    bn._omnihd_pending_batches = 0


def _state_dict_pre_hook(mod, prefix, keep_vars):          # module-level functions: a model carrying the hooks pickles
    _flush_batches_tracked(mod)


def _load_state_dict_post_hook(mod, incompatible):
    mod._omnihd_pending_batches = 0


def flush_batch_counters(model):
    """Write every BatchNorm's host-side batch count into its ``num_batches_tracked`` buffer.  ``state_dict()`` and leaving
    training mode through ``bn_act`` do this by themselves; call it before anything that reads the buffers WITHOUT going
    through ``nn.Module.state_dict`` — mmcv's ``get_state_dict`` walks ``_save_to_state_dict`` directly, ``torch.save(model)``
    pickles the buffers, DDP with ``broadcast_buffers=True`` sends them."""
    n = 0
    for m in model.modules():
        if getattr(m, "_omnihd_pending_batches", 0):
            _flush_batches_tracked(m)
            n += 1
    return n


def _count_batch(bn):
    """``num_batches_tracked += 1`` without a kernel per layer and step (37 launches per step in the fusion detector): the
    fused path uses a fixed momentum and never reads the counter, so increments are kept on the host and written into the
    buffer when it is needed — before ``state_dict()`` (checkpoints carry the same value torch would), when the layer runs in
    eval mode or through ``bn_act``'s plain branch, and by ``flush_batch_counters(model)`` for checkpoint writers that bypass
    ``nn.Module.state_dict`` (see there)."""
    if not hasattr(bn, "_omnihd_pending_batches"):
        bn._omnihd_pending_batches = 0
        bn.register_state_dict_pre_hook(_state_dict_pre_hook)
        bn.register_load_state_dict_post_hook(_load_state_dict_post_hook)
    bn._omnihd_pending_batches += 1


def _bn_train_fused_applies(channels, bn):
    """Will ``bn_act`` take its fused training branch on ONE rank for an fp32 (N, channels, H, W) device tensor?  (What the
    planes-only gradient contract needs to know before the convolution runs; mirrors ops.bn_train_supported.)"""
    if not (isinstance(bn, nn.modules.batchnorm._BatchNorm) and bn.training and bn.affine and bn.momentum is not None
            and bn.track_running_stats and channels % 8 == 0 and channels <= 2048):
        return False
    if getattr(bn, "_omnihd_sync", False) or isinstance(bn, nn.SyncBatchNorm):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return False                              # the exchange form of the backward has no plane output (yet)
    return True


def _hooked(*mods):
    return any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks for m in mods)


def conv_bn_act(conv, bn, x, relu=True, residual=None):
    """``relu(bn(conv(x)))`` for a convolution directly followed by its BatchNorm (nothing else sees the tensor between them).
    fp32 step, split kernels in both backward directions of the convolution, fused training BatchNorm: the gradient between
    the two travels as bf16 planes ONLY — the BatchNorm backward writes hi / lo instead of the fp32 tensor, the convolution's
    backward reads them, and the split pass over that gradient (one launch + a read and a write of the whole tensor per layer)
    is gone.  Everything else: the plain composition ``bn_act(conv(x), bn)``."""
    from .. import ops
    if (isinstance(conv, BevConv2d) and not _hooked(conv, bn) and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
            and torch.is_grad_enabled() and conv.groups == 1 and conv.padding_mode == "zeros" and not isinstance(conv.padding, str)
            and x.dim() == 4 and ops._fp32_policy() != "miopen"
            and ops.conv_split_supported(x, conv.weight, conv.stride, conv.padding, conv.dilation)
            and not ops.conv_split_all_miopen(x.shape, conv.weight.shape[0], conv.weight.shape[2], conv.stride, conv.padding,
                                              conv.dilation, x.device.index)
            and x.numel() > 0 and _bn_train_fused_applies(conv.out_channels, bn)
            and (residual is None or (residual.is_cuda and residual.dtype == torch.float32))
            and ops.conv_grad_planes_ok(x.shape, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, x.device.index)):
        y = ops.conv_split(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, grad_planes_only=True)
        if residual is None or (residual.shape == y.shape and residual.dtype == y.dtype):
            return bn_act(y, bn, relu=relu, residual=residual, _grad_planes_only=True)
        raise RuntimeError("conv_bn_act: the residual does not have the convolution's output shape / dtype")
    return bn_act(conv(x), bn, relu=relu, residual=residual)


def bn_act(x, bn, relu=True, residual=None, inplace=True, _grad_planes_only=False):
    """``relu(bn(x) + residual)``.  A frozen BatchNorm (eval mode, affine parameters without gradient) on a bf16 or fp32
    device tensor is an affine map with constant coefficients: ONE fused channels-last pass each way
    (csrc/affine_act.hip).  A training-mode BatchNorm on a device tensor runs on csrc/batch_norm.hip in either precision
    (statistics exchanged between ranks for naiveSyncBN and torch SyncBatchNorm layers).  Everything else — CPU tensors,
    channel counts that are not multiples of 8, eval-mode layers with trainable affine parameters — is the plain torch
    composition."""
    from .. import ops
    if not bn.training and getattr(bn, "_omnihd_pending_batches", 0):
        _flush_batches_tracked(bn)                # the layer has left training mode: its counter buffer is made current
    if (isinstance(bn, nn.modules.batchnorm._BatchNorm) and not bn.training and bn.track_running_stats and bn.affine
            and not bn.weight.requires_grad and not bn.bias.requires_grad and ops.affine_act_supported(x, residual)):
        scale, shift = frozen_bn_constants(bn)
        return ops.affine_act(x, scale, shift, residual, relu)
    if (isinstance(bn, nn.modules.batchnorm._BatchNorm) and bn.training and bn.affine and bn.momentum is not None
            and bn.track_running_stats and ops.bn_train_supported(x)
            and (residual is None or (residual.shape == x.shape and residual.dtype == x.dtype))):
        group, unbiased_sync = None, False
        torch_sync = isinstance(bn, nn.SyncBatchNorm)
        if getattr(bn, "_omnihd_sync", False) or torch_sync:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                g = (bn.process_group if torch_sync and bn.process_group is not None else dist.group.WORLD)
                if dist.get_world_size(g) > 1:
                    group = g
        if torch_sync and group is not None:
            # torch's SyncBatchNorm weights every rank by its row count; the fused exchange averages per-rank moments, which
            # is the same thing only when all ranks hold the same number of rows: image-shaped (4-D) activations.  Anything
            # else (per-rank pillar counts) keeps torch's own implementation.
            if x.dim() != 4:
                out = bn(x)
                if residual is not None:
                    out = out + residual
                return F.relu(out, inplace=inplace) if relu else out
            unbiased_sync = True
        if bn.num_batches_tracked is not None:
            _count_batch(bn)
        return ops.bn_train_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, relu, group,
                                residual, unbiased_sync, grad_planes_only=_grad_planes_only and group is None)
    if getattr(bn, "_omnihd_pending_batches", 0):
        _flush_batches_tracked(bn)                # torch's own forward may read the counter (momentum=None)
    out = bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out, inplace=inplace) if relu else out


def run_fused(seq, x):
    """Run an nn.Sequential, executing every (BatchNorm, ReLU) pair as one ``bn_act`` call."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, BevConv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.modules.batchnorm._BatchNorm):
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            x = conv_bn_act(m, mods[i + 1], x, relu=relu)
            i += 3 if relu else 2
            continue
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = bn_act(x, m, relu=relu)
            i += 2 if relu else 1
        else:
            x = m(x)
            i += 1
    return x


class BevConv2d(nn.Conv2d):
    """nn.Conv2d (same parameters / state-dict keys) whose bf16 training path runs on the hand-written MFMA kernels of this
    library where they measure faster than MIOpen for the layer's geometry: forward and data gradient on the implicit-GEMM
    kernel (omnihd_conv_fwd_bf16), weight gradient on the k-major chain (omnihd_conv_wgrad_bf16)."""

    def _conv_forward(self, x, weight, bias):
        from .. import ops
        if (self.training and weight.requires_grad and torch.is_autocast_enabled() and x.is_cuda and self.groups == 1
                and self.padding_mode == "zeros" and not isinstance(self.padding, str)):
            # once a geometry has been measured in MIOpen's favour the layer is a plain convolution again
            # (no Python in its backward); unmeasured geometries go through the function that measures
            if ops.conv_all_miopen(x.shape, weight.shape[0], weight.shape[2], self.stride[0], self.padding[0],
                                   self.dilation[0], x.device.index):
                # MIOpen in all three directions: a plain convolution, fed with the bf16 image of the master weight that
                # the step refreshes with one fused copy (no per-layer cast kernel of autocast)
                return super()._conv_forward(x, ops.bf16_weight(weight), bias)
            xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            if ops.conv_wgrad_supported(xb, weight, self.stride, self.padding, self.dilation):
                return ops.conv_hip_wgrad(xb, weight, bias, self.stride, self.padding, self.dilation)
        if (x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled() and self.groups == 1
                and self.padding_mode == "zeros" and not isinstance(self.padding, str)
                and _env("OMNIHD_FP32_CONV", "tune") != "miopen"
                and ops.conv_split_supported(x, weight, self.stride, self.padding, self.dilation)
                and not ops.conv_split_all_miopen(x.shape, weight.shape[0], weight.shape[2], self.stride, self.padding,
                                                  self.dilation, x.device.index)):
            # the reference-precision (fp32) step: fp32-grade result from three bf16 MFMA products per term
            # (csrc/conv_igemm.hip, SPLIT), where that measures faster than MIOpen's fp32 kernel for the geometry
            return ops.conv_split(x, weight, bias, self.stride, self.padding, self.dilation)
        if (bias is not None and weight.shape[0] % 8 != 0 and x.is_cuda and x.dim() == 4 and self.padding_mode == "zeros"
                and not isinstance(self.padding, str) and bias.requires_grad and torch.is_grad_enabled()):
            # odd channel counts (DepthNet's 59 depth logits): bias gradient from the column-sum kernel instead of torch's
            # element-wise NHWC reduction (0.36 ms -> 10 us).  The convolution runs inside the function (fresh output tensor);
            # under autocast the operands are cast here as autocast would cast them
            xe, we = x, weight
            if torch.is_autocast_enabled():
                dt = torch.get_autocast_dtype("cuda")
                xe = x.to(dt)
                we = ops.bf16_weight(weight) if dt == torch.bfloat16 else weight.to(dt)
            if ops.conv_bias_colsum_supported(xe, we, bias):
                return ops.conv_bias_colsum(xe, we, bias, self.stride, self.padding, self.dilation, self.groups)
        return super()._conv_forward(x, weight, bias)


class BevConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d with kernel == stride (SECONDFPN's up-sampling blocks): weight gradient as a 1x1
    weight gradient on the MFMA kernel; everything else unchanged."""

    def forward(self, x, output_size=None):
        from .. import ops
        if (self.training and self.weight.requires_grad and torch.is_autocast_enabled() and output_size is None):
            xb = x.to(torch.bfloat16) if x.is_cuda else x
            if ops.deconv_supported(xb, self.weight, self.kernel_size, self.stride, self.padding, self.output_padding,
                                    self.groups, self.dilation, self.bias):
                return ops.deconv_hip_wgrad(xb.contiguous(memory_format=torch.channels_last), self.weight,
                                            self.kernel_size[0])
        if (output_size is None and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()
                and ops.deconv_split_supported(x, self.weight, self.kernel_size, self.stride, self.padding, self.output_padding,
                                               self.groups, self.dilation, self.bias)
                and (ops.deterministic() or _env("OMNIHD_DECONV_SPLIT", "1") != "0")):
            # the reference-precision (fp32) step: forward / data gradient on the general implicit-GEMM kernel in split form
            return ops.deconv_split(x, self.weight, self.kernel_size[0])
        return super().forward(x, output_size)


def use_bev_conv(module):
    """Switch every dense nn.Conv2d (square 1x1 / 3x3 kernel, groups 1, channel counts multiples of 8) and every
    nn.ConvTranspose2d with kernel == stride under ``module`` to the classes above, in place; parameters and
    state-dict keys are kept.  Shapes the kernel does not take at run time simply use the MIOpen path."""
    n = 0
    for m in module.modules():
        if type(m) is nn.Conv2d and m.groups == 1 and m.kernel_size in ((1, 1), (3, 3)) \
                and m.in_channels % 8 == 0 and (m.out_channels % 8 == 0 or m.bias is not None):
            # (an output channel count that is not a multiple of 8 stays on MIOpen; the class then only takes over the
            # bias gradient, which torch reduces element by element for such widths)
            m.__class__ = BevConv2d
            n += 1
        elif type(m) is nn.ConvTranspose2d and m.groups == 1 and m.kernel_size == m.stride and m.bias is None:
            m.__class__ = BevConvTranspose2d
            n += 1
    return n



class BilinearResize(nn.Module):
    """``nn.Upsample(size, mode='bilinear', align_corners=True)`` as two small matrix products
    (out = Rh @ x @ Rw^T; bilinear resampling is separable and linear).  Same values up to fp rounding;
    forward AND backward are GEMMs instead of the slow gather/scatter interpolation kernels."""

    def __init__(self, size):
        super().__init__()
        self.size = tuple(size)
        self._cache = {}

    @staticmethod
    def _matrix(n_out, n_in, device, dtype):
        m = torch.zeros(n_out, n_in, device=device, dtype=torch.float32)
        if n_in == 1 or n_out == 1:
            m[:, 0] = 1.0
            return m.to(dtype)
        pos = torch.arange(n_out, device=device, dtype=torch.float32) * ((n_in - 1) / (n_out - 1))
        lo = pos.floor().clamp(max=n_in - 1).long()
        hi = (lo + 1).clamp(max=n_in - 1)
        w = pos - lo.float()
        idx = torch.arange(n_out, device=device)
        m[idx, lo] += 1 - w
        m[idx, hi] += w
        return m.to(dtype)

    def forward(self, x):
        H, W = x.shape[-2:]
        key = (H, W, str(x.device), x.dtype)
        if key not in self._cache:
            self._cache[key] = (self._matrix(self.size[0], H, x.device, x.dtype), self._matrix(self.size[1], W, x.device, x.dtype))
        rh, rw = self._cache[key]
        if (x.dim() == 4 and x.is_cuda and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
                and _env("OMNIHD_RESIZE_CL", "1") != "0"):
            # channels_last in, channels_last out (round 5): the NCHW product left a contiguous NCHW tensor in front of the NHWC
            # convolution kernels — a layout copy forward and mixed-layout gradient sums backward per pyramid level
            y = torch.einsum("ih,bhwc,jw->bijc", rh, x.permute(0, 2, 3, 1), rw)
            return y.permute(0, 3, 1, 2)
        return torch.einsum("ih,bchw,jw->bcij", rh, x, rw)


class ConvModule(nn.Module):
    """conv -> norm -> act with mmcv's attribute names (``.conv``, ``.bn``, ``.activate``) so that
    state-dict keys such as ``reduc_conv.conv.weight`` / ``reduc_conv.bn.weight`` match."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias="auto", conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == "auto":
            bias = not self.with_norm
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            a = dict(act_cfg)
            if a["type"] == "ReLU":
                a.setdefault("inplace", inplace)
            self.activate = build_activation(a)
        nn.init.kaiming_normal_(self.conv.weight, mode="fan_out", nonlinearity="relu")
        if getattr(self.conv, "bias", None) is not None:
            nn.init.zeros_(self.conv.bias)

    @property
    def norm(self):
        return getattr(self, self.norm_name) if self.with_norm else None

    def forward(self, x):
        relu = self.with_activation and isinstance(self.activate, nn.ReLU)
        if self.with_norm and isinstance(self.norm, nn.modules.batchnorm._BatchNorm) and isinstance(self.conv, BevConv2d):
            x = conv_bn_act(self.conv, self.norm, x, relu=relu)       # the tensor between conv and norm is seen by nobody else
            if self.with_activation and not relu:
                x = self.activate(x)
            return x
        x = self.conv(x)
        if self.with_norm:
            x = bn_act(x, self.norm, relu=relu) if isinstance(self.norm, nn.modules.batchnorm._BatchNorm) else self.norm(x)
            if self.with_activation and not (relu and isinstance(self.norm, nn.modules.batchnorm._BatchNorm)):
                x = self.activate(x)
        elif self.with_activation:
            x = self.activate(x)
        return x
