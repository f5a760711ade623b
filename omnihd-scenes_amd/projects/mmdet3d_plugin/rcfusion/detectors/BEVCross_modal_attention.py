"""Cross-modal spatial attention of RCFusion — mirror of
projects/mmdet3d_plugin/rcfusion/detectors/BEVCross_modal_attention.py:6-43 (same class name, constructor,
sub-module names ``att_img.0``, ``att_radar.0``, ``reduce_mixBEV.conv/bn``).  Each modality is gated by
the OTHER modality's spatial attention map (sigmoid of a 3x3 conv over [channel-mean, channel-max]),
then both are concatenated and reduced 640 -> 384."""
import torch
from torch import nn

from omnihd_amd.mm import ConvModule

__all__ = ["Cross_Modal_Fusion"]


class Cross_Modal_Fusion(nn.Module):
    def __init__(self, kernel_size=3, norm_cfg=None):
        super().__init__()
        assert kernel_size in (3, 7), "kernel size must be 3 or 7"
        padding = 3 if kernel_size == 7 else 1
        self.att_img = nn.Sequential(nn.Conv2d(2, 1, kernel_size, padding=padding, bias=False), nn.Sigmoid())
        self.att_radar = nn.Sequential(nn.Conv2d(2, 1, kernel_size, padding=padding, bias=False), nn.Sigmoid())
        self.reduce_mixBEV = ConvModule(256 + 384, 384, 3, padding=1, conv_cfg=None, norm_cfg=norm_cfg,
                                        act_cfg=dict(type="ReLU"), inplace=False)

    @staticmethod
    def _descriptor(x):
        return torch.cat([x.mean(dim=1, keepdim=True), x.max(dim=1, keepdim=True)[0]], dim=1)

    def forward(self, img_bev, radar_bev):
        img_att = self.att_img(self._descriptor(img_bev))
        radar_att = self.att_radar(self._descriptor(radar_bev))
        fused = torch.cat([img_bev * radar_att, radar_bev * img_att], dim=1)
        return self.reduce_mixBEV(fused)
