"""``FPNC`` image neck: FPN, every level resized to (H/downsample, W/downsample), a 1x1 adapter per
level, concat, 3x3 reduce conv.  Mirrors projects/mmdet3d_plugin/bevfusion/necks/fpnc.py:45-118
(constructor arguments, sub-module names ``adp.{i}.1.conv``, ``reduc_conv.conv``).  Dense convs run
on MIOpen through torch."""
import torch
import torch.nn.functional as F
from torch import nn

from omnihd_amd.mm import NECKS, ConvModule
from omnihd_amd.mm.bricks import BilinearResize
from omnihd_amd.mm.fpn import FPN

__all__ = ["FPNC"]


@NECKS.register_module()
class FPNC(FPN):
    def __init__(self, conv_cfg=None, norm_cfg=None, act_cfg=None, final_dim=(900, 1600), downsample=4,
                 use_adp=False, fuse_conv_cfg=None, outC=256, **kwargs):
        super().__init__(conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg, **kwargs)
        self.target_size = (final_dim[0] // downsample, final_dim[1] // downsample)
        self.use_adp = use_adp
        if use_adp:
            adp = []
            for i in range(self.num_outs):
                # (the reference uses nn.Upsample(bilinear, align_corners=True): same map, as two GEMMs)
                resize = (nn.AdaptiveAvgPool2d(self.target_size) if i == 0 else BilinearResize(self.target_size))
                adp.append(nn.Sequential(resize, ConvModule(self.out_channels, self.out_channels, 1, padding=0,
                                                            conv_cfg=fuse_conv_cfg, norm_cfg=norm_cfg,
                                                            act_cfg=act_cfg, inplace=False)))
            self.adp = nn.ModuleList(adp)
        self.reduc_conv = ConvModule(self.out_channels * self.num_outs, outC, 3, padding=1, conv_cfg=fuse_conv_cfg,
                                     norm_cfg=norm_cfg, act_cfg=act_cfg, inplace=False)

    def forward(self, x):
        outs = super().forward(x)
        if len(outs) == 1:
            return [outs[0]]
        if self.use_adp:
            resized = [self.adp[i](o) for i, o in enumerate(outs)]
        else:
            resized = [o if o.shape[2:] == self.target_size else
                       F.interpolate(o, self.target_size, mode="bilinear", align_corners=True) for o in outs]
        return [self.reduc_conv(torch.cat(resized, dim=1))]
