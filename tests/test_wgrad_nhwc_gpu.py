"""GPU parity of the NHWC weight-gradient kernel (csrc/conv_wgrad_nhwc.hip: tiles read transposed from LDS, no staging pass) against
torch's fp32 weight gradient: split form on fp32 operands 1e-4 of the largest entry, bf16 form on the bf16-rounded operands 2e-3;
strides, paddings, dilations, ragged channel counts, pixel counts that are not multiples of the K-step, split-K and single-split
launches; results run-to-run identical.  Layers: ResNet-50 / FPN / DepthNet /
SECOND / head of the reference config (projects/configs/bevfusion_NewScenes/bevfusion.py:62-123)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


GEOMS = [  # B, H, W, cin, cout, k, stride, pad, dil
    (6, 64, 176, 256, 64, 1, 1, 0, 1),        # ResNet bottleneck conv1
    (6, 64, 176, 64, 64, 3, 1, 1, 1),         # bottleneck conv2 (layer1)
    (6, 64, 176, 128, 128, 3, 2, 1, 1),       # layer2.0.conv2, stride 2
    (6, 64, 176, 256, 512, 1, 2, 0, 1),       # layer2.0.downsample
    (6, 16, 44, 1024, 256, 1, 1, 0, 1),       # FPN lateral
    (6, 8, 22, 512, 2048, 1, 1, 0, 1),        # layer4 conv3
    (1, 160, 240, 64, 64, 3, 1, 1, 1),        # SECOND stage 0
    (1, 40, 60, 256, 256, 3, 1, 1, 1),        # SECOND stage 2
    (1, 160, 240, 384, 72, 1, 1, 0, 1),       # head regression branch
    (2, 33, 51, 72, 40, 3, 2, 1, 1),          # odd sizes, ragged channel tiles
    (1, 30, 41, 64, 24, 3, 1, 2, 2),          # dilation 2
    (1, 9, 7, 8, 8, 3, 1, 1, 1),              # 63 pixels: one short split
    (2, 16, 20, 96, 48, 2, 2, 0, 1),          # kernel == stride 2
    (6, 64, 176, 256, 256, 3, 1, 1, 1),       # DepthNet 3x3: three-taps form, 2 x 2 channel tiles
    (2, 33, 51, 72, 40, 3, 1, 1, 1),          # three taps, odd sizes, ragged channel tiles
    (6, 8, 22, 512, 512, 3, 1, 1, 1),         # three taps, image rows shorter than a K-step
    (3, 5, 4, 16, 24, 3, 1, 1, 1),            # three taps, rows of 4 pixels
    (1, 2, 3, 8, 8, 3, 1, 1, 1),              # three taps, six pixels
]


@pytest.mark.parametrize("B,H,W,cin,cout,k,s,p,d", GEOMS)
def test_split_and_bf16_forms_match_the_fp32_weight_gradient(cuda, B, H, W, cin, cout, k, s, p, d, monkeypatch):
    from omnihd_amd import ops
    torch.manual_seed(B * 100 + H + cin + cout + k)
    x = torch.randn(B, cin, H, W, device=cuda).contiguous(memory_format=torch.channels_last)
    Ho, Wo = (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1
    g = torch.randn(B, cout, Ho, Wo, device=cuda).contiguous(memory_format=torch.channels_last)
    want = torch.nn.grad.conv2d_weight(x, (cout, cin, k, k), g, stride=s, padding=p, dilation=d)
    assert ops.wgrad_nhwc_preferred(B, H, W, cin, Ho, Wo, cout, k, s, p, d)
    got = ops.conv_wgrad_split(ops.split_f32(x), ops.split_f32(g), k, s, p, d)
    assert got.shape == want.shape and got.dtype == torch.float32
    assert _rel(got, want) <= 1e-4, _rel(got, want)
    assert torch.equal(got, ops.conv_wgrad_split(ops.split_f32(x), ops.split_f32(g), k, s, p, d))     # fixed-order slab sum
    xb, gb = x.to(torch.bfloat16), g.to(torch.bfloat16)
    want_b = torch.nn.grad.conv2d_weight(xb.float(), (cout, cin, k, k), gb.float(), stride=s, padding=p, dilation=d)
    got_b = ops.conv_wgrad(xb, gb, k, s, p, d)
    assert _rel(got_b, want_b) <= 2e-3, _rel(got_b, want_b)


def test_the_routing_rule_takes_every_detector_geometry_and_leaves_the_rest_to_the_library(cuda):
    from omnihd_amd import ops
    assert ops.wgrad_nhwc_preferred(1, 160, 240, 1024, 160, 240, 1024, 3, 1, 1, 1)          # BEV encoder (three-taps form)
    assert ops.wgrad_nhwc_preferred(6, 64, 176, 256, 64, 176, 256, 3, 1, 1, 1)              # DepthNet 3x3 at 6 x 64 x 176
    assert ops.wgrad_nhwc_preferred(1, 160, 240, 64, 160, 240, 64, 3, 1, 1, 1)              # SECOND stage 0
    assert ops.wgrad_nhwc_preferred(6, 64, 176, 256, 64, 176, 64, 1, 1, 0, 1)               # 1x1
    assert ops.wgrad_nhwc_preferred(6, 64, 176, 128, 32, 88, 128, 3, 2, 1, 1)               # strided
    assert not ops.wgrad_nhwc_preferred(1, 64, 64, 64, 64, 64, 64, 5, 1, 2, 1)              # 5x5: not taken
    assert not ops.wgrad_nhwc_preferred(1, 64, 64, 60, 64, 64, 64, 3, 1, 1, 1)              # channels not a multiple of 8
    x = torch.zeros(1, 64, 64, 64, device=cuda, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with pytest.raises(ValueError, match="does not take"):                                  # no silent second path: the caller routes to MIOpen
        ops.conv_wgrad(x, x, 5, 1, 2, 1)
    assert not ops.conv_split_geometry((1, 60, 64, 64), 64, 3, (1, 1), (1, 1), (1, 1))[2]
