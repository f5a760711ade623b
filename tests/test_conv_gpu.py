"""GPU parity of the hand-written MFMA weight-gradient kernel against torch's fp32 convolution backward
on the same bf16-rounded inputs (only the accumulation order differs: tolerance 2e-3 of the gradient's
max magnitude, bf16 products are exact in fp32)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def ref_wgrad(x, g):
    w = torch.zeros(g.shape[1], x.shape[1], 3, 3, device=x.device, requires_grad=True)
    y = F.conv2d(x.float(), w, padding=1)
    return torch.autograd.grad(y, w, g.float())[0]


@pytest.mark.parametrize("B,H,W,cin,cout", [(1, 16, 24, 128, 128), (2, 9, 40, 256, 128), (1, 160, 240, 128, 256), (1, 7, 8, 384, 640)])
def test_wgrad_matches_fp32_reference(cuda, B, H, W, cin, cout):
    from omnihd_amd import ops
    torch.manual_seed(B * H + cin)
    x = torch.randn(B, cin, H, W, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = (torch.randn(B, cout, H, W, device=cuda) * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    got = ops.conv3x3_wgrad(x, g)
    assert got.shape == (cout, cin, 3, 3) and got.dtype == torch.float32
    want = ref_wgrad(x, g)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-3 * scale
    assert torch.equal(got, ops.conv3x3_wgrad(x, g))          # deterministic


def test_border_taps_see_zero_padding(cuda):
    """x = 1 everywhere, g = 1 everywhere: dW[tap] = number of pixels whose shifted neighbour is inside."""
    from omnihd_amd import ops
    B, H, W, C = 1, 8, 16, 128
    x = torch.ones(B, C, H, W, device=cuda, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.ones(B, C, H, W, device=cuda, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = ops.conv3x3_wgrad(x, g)
    want = torch.tensor([[(H - abs(dy)) * (W - abs(dx)) for dx in (-1, 0, 1)] for dy in (-1, 0, 1)], dtype=torch.float32, device=cuda)
    assert torch.equal(dw[5, 77], want) and torch.equal(dw[127, 0], want)


def test_bev_conv_module_trains_like_nn_conv2d(cuda):
    from omnihd_amd.mm.bricks import BevConv2d, use_bev_conv
    torch.manual_seed(0)
    seq = torch.nn.Sequential(torch.nn.Conv2d(128, 256, 3, padding=1, bias=False), torch.nn.ReLU(),
                              torch.nn.Conv2d(256, 128, 3, padding=1, bias=False)).to(cuda).to(memory_format=torch.channels_last)
    ref = torch.nn.Sequential(torch.nn.Conv2d(128, 256, 3, padding=1, bias=False), torch.nn.ReLU(),
                              torch.nn.Conv2d(256, 128, 3, padding=1, bias=False)).to(cuda).to(memory_format=torch.channels_last)
    ref.load_state_dict(seq.state_dict())
    use_bev_conv(seq)
    assert isinstance(seq[0], BevConv2d) and list(seq.state_dict()) == list(ref.state_dict())
    x = torch.randn(2, 128, 20, 24, device=cuda).contiguous(memory_format=torch.channels_last)
    for m in (seq, ref):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            m(x).float().square().mean().backward()
    for a, b in zip(seq.parameters(), ref.parameters()):
        assert a.grad.dtype == torch.float32
        torch.testing.assert_close(a.grad, b.grad, rtol=3e-2, atol=3e-2 * float(b.grad.abs().max()))


def test_dcn_hip_sampling_matches_torch_formulation(cuda):
    """HIP deformable sampling (fwd, grad wrt input, grad wrt offsets) vs the embedding_bag formulation in fp32 on
    the same bf16-rounded inputs; offsets reach beyond one pixel and outside the image."""
    from omnihd_amd.mm.dcn import DeformConv2dPack
    torch.manual_seed(0)
    m = DeformConv2dPack(64, 64, 3, padding=1, groups=4).to(cuda)
    torch.nn.init.normal_(m.conv_offset.weight, std=0.05)
    torch.nn.init.normal_(m.conv_offset.bias, std=1.2)
    x = torch.randn(2, 64, 12, 20, device=cuda).to(torch.bfloat16).float().contiguous(memory_format=torch.channels_last).requires_grad_()
    off = m.conv_offset(x).detach().requires_grad_()
    a = m._hip_sample_and_gemm(x, off)
    b = m._gather_and_gemm(x, off, torch.float32)
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2e-2 * scale
    g = torch.randn_like(b)
    ga = torch.autograd.grad(a, [x, off, m.weight], g, retain_graph=True)
    gb = torch.autograd.grad(b, [x, off, m.weight], g)
    for u, v, name in zip(ga, gb, ("x", "offset", "weight")):
        assert float((u - v).abs().max()) <= 3e-2 * float(v.abs().max()), name
    again = m._hip_sample_and_gemm(x, off)
    assert torch.equal(a, again)


@pytest.mark.parametrize("B,H,W,cin,cout,stride", [(2, 8, 22, 512, 128, 1), (6, 16, 44, 256, 1024, 1), (1, 9, 13, 128, 256, 2), (3, 7, 5, 2048, 512, 1)])
def test_wgrad_1x1_matches_fp32_reference(cuda, B, H, W, cin, cout, stride):
    from omnihd_amd import ops
    torch.manual_seed(cin + H)
    x = torch.randn(B, cin, H, W, device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device=cuda) * 0.05).to(torch.bfloat16).requires_grad_()
    y = ops.conv_hip_wgrad(x, w, None, (stride, stride), (0, 0))
    g = (torch.randn_like(y.float()) * 0.1).to(torch.bfloat16)
    (gw,) = torch.autograd.grad(y, w, g)
    wr = w.detach().float().requires_grad_()
    yr = F.conv2d(x.float(), wr, None, stride)
    (want,) = torch.autograd.grad(yr, wr, g.float())
    assert gw.shape == want.shape
    assert float((gw.float() - want).abs().max()) <= 1e-2 * float(want.abs().max())      # bf16 output rounding
