"""ResNet backbone + BasicBlock/Bottleneck with mmdet 2.14 attribute names (``conv1``, ``bn1``,
``layer{1..4}.{i}.conv{j}``, ``downsample.{0,1}``) — used by the reference through
``img_backbone=dict(type='ResNet', depth=50, ...)`` (projects/configs/bevfusion_NewScenes/
bevfusion.py:76-85) and ``BasicBlock`` inside DepthNet (cam_stream_lss_bevpoolv2_depthnet.py:583-585).
Dense convolutions: executed by MIOpen through torch (channels-last bf16 when the harness enables it)."""
from torch import nn

from .bricks import conv_bn_act, bn_act, build_norm_layer
from .registry import BACKBONES


def _downsample(seq, x):
    """``downsample`` is Sequential(conv, norm) (mmdet naming downsample.0 / downsample.1)."""
    if isinstance(seq, nn.Sequential) and len(seq) == 2:
        return conv_bn_act(seq[0], seq[1], x, relu=False)
    return seq(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, norm_cfg=dict(type="BN"), **_):
        super().__init__()
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = conv_bn_act(self.conv1, getattr(self, self.norm1_name), x)
        if self.downsample is not None:
            identity = _downsample(self.downsample, x)
        return conv_bn_act(self.conv2, getattr(self, self.norm2_name), out, residual=identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch",
                 norm_cfg=dict(type="BN"), **_):
        super().__init__()
        s1, s2 = (1, stride) if style == "pytorch" else (stride, 1)
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.norm3_name, norm3 = build_norm_layer(norm_cfg, planes * 4, postfix=3)
        self.conv1 = nn.Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=s2, padding=dilation, dilation=dilation, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.add_module(self.norm3_name, norm3)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = conv_bn_act(self.conv1, getattr(self, self.norm1_name), x)
        out = conv_bn_act(self.conv2, getattr(self, self.norm2_name), out)
        if self.downsample is not None:
            identity = _downsample(self.downsample, x)
        return conv_bn_act(self.conv3, getattr(self, self.norm3_name), out, residual=identity)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)),
                     50: (Bottleneck, (3, 4, 6, 3)), 101: (Bottleneck, (3, 4, 23, 3))}

    def __init__(self, depth, in_channels=3, base_channels=64, num_stages=4, strides=(1, 2, 2, 2),
                 dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style="pytorch", frozen_stages=-1,
                 norm_cfg=dict(type="BN", requires_grad=True), norm_eval=True, **_):
        super().__init__()
        block, stage_blocks = self.arch_settings[depth]
        self.out_indices, self.frozen_stages, self.norm_eval = out_indices, frozen_stages, norm_eval
        self.conv1 = nn.Conv2d(in_channels, base_channels, 7, stride=2, padding=3, bias=False)
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, base_channels, postfix=1)
        self.add_module(self.norm1_name, norm1)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = base_channels
        self.res_layers = []
        for i in range(num_stages):
            planes = base_channels * 2 ** i
            layers = []
            for j in range(stage_blocks[i]):
                stride = strides[i] if j == 0 else 1
                down = None
                if j == 0 and (stride != 1 or inplanes != planes * block.expansion):
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                         build_norm_layer(norm_cfg, planes * block.expansion)[1])
                layers.append(block(inplanes, planes, stride=stride, dilation=dilations[i], downsample=down,
                                    style=style, norm_cfg=norm_cfg))
                inplanes = planes * block.expansion
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            getattr(self, self.norm1_name).eval()
            for m in (self.conv1, getattr(self, self.norm1_name)):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f"layer{i}")
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def forward(self, x):
        x = self.maxpool(bn_act(self.conv1(x), getattr(self, self.norm1_name)))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self
