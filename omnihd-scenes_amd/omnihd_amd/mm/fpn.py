"""mmdet 2.14 ``FPN`` (lateral_convs / fpn_convs of ConvModules, top-down nearest upsampling,
extra levels by stride-2 max-pool when ``add_extra_convs`` is False) — base class of the plugin's
``FPNC`` neck (projects/mmdet3d_plugin/bevfusion/necks/fpnc.py:45)."""
import torch.nn.functional as F
from torch import nn

from .bricks import ConvModule
from .registry import NECKS


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None,
                 act_cfg=None, upsample_cfg=dict(mode="nearest"), **_):
        super().__init__()
        assert not add_extra_convs, "only the configuration used by the reference configs is restated"
        self.in_channels, self.out_channels, self.num_outs = in_channels, out_channels, num_outs
        self.num_ins = len(in_channels)
        self.backbone_end_level = self.num_ins if end_level == -1 else end_level
        self.start_level = start_level
        self.upsample_cfg = dict(upsample_cfg)
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=None if no_norm_on_lateral else norm_cfg,
                                                 act_cfg=act_cfg, inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg,
                                             norm_cfg=norm_cfg, act_cfg=act_cfg, inplace=False))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)

    def forward(self, inputs):
        laterals = [l(inputs[i + self.start_level]) for i, l in enumerate(self.lateral_convs)]
        for i in range(len(laterals) - 1, 0, -1):
            laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], size=laterals[i - 1].shape[2:],
                                                              **self.upsample_cfg)
        outs = [self.fpn_convs[i](laterals[i]) for i in range(len(laterals))]
        for _ in range(self.num_outs - len(outs)):
            outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return tuple(outs)
