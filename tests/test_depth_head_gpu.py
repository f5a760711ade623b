"""GPU parity tests of the depth-head epilogue kernel (csrc/depth_head.hip: softmax over D + depth / context split + the
pooling's layouts, reference cam_stream_lss_bevpoolv2_depthnet.py:134-143, :290) and of the pooling forward that keeps the
empty rows of its output buffer between launches of one plan.

Tolerances: distribution vs torch.softmax in fp32 <= 1e-6 absolute (fp32 exp / sum in another order); the context rows and
both layouts bit-exact copies; gradients vs autograd of the torch formulation 1e-5 relative; epilogue + HIP pooling vs
torch.softmax + the CPU oracle pooling 1e-5 relative (north_star allows 1e-3); kept-rows forward bit-identical to a
fresh-buffer forward."""
import numpy as np
import pytest
import torch

from oracle import cpu as OC
from oracle import lss_oracle as O
from tests.helpers import full_size_geometry, t

pytestmark = pytest.mark.gpu


def _reference(logits, context):
    """The reference's formulation: x[:, :D].softmax(1) in fp32, feat = context permuted to pixel rows."""
    depth = logits.float().softmax(dim=1)
    return depth, depth.permute(0, 2, 3, 1).contiguous(), context.float().permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,D,C,H,W", [(6, 59, 64, 64, 176), (3, 8, 64, 8, 12), (2, 59, 64, 5, 13), (1, 160, 8, 3, 5)])
def test_depth_head_forward_matches_torch_softmax_and_layouts(cuda, dtype, M, D, C, H, W):
    from omnihd_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 1000 + D)
    logits = (torch.randn(M, D, H, W, generator=g) * 3).to(cuda).to(dtype).contiguous(memory_format=torch.channels_last)
    context = torch.randn(M, C, H, W, generator=g).to(cuda).to(dtype).contiguous(memory_format=torch.channels_last)
    depth, rows, feat = ops.depth_head(logits, context, want_rows=True)
    wd, wr, wf = _reference(logits, context)
    assert depth.shape == (M, D, H, W) and depth.is_contiguous() and depth.dtype == torch.float32
    assert rows.shape == (M, H, W, D) and feat.shape == (M, H, W, C) and feat.dtype == torch.float32
    assert float((depth - wd).abs().max()) <= 1e-6
    assert torch.equal(rows, depth.permute(0, 2, 3, 1))                  # the two layouts hold the same bits
    assert torch.equal(feat.contiguous(), wf)                            # the context rows are a copy (exact, also from bf16)
    assert float((depth.sum(1) - 1).abs().max()) <= 1e-5
    _, none_rows, _ = ops.depth_head(logits, context, want_rows=False)
    assert none_rows is None


def test_depth_head_takes_channel_slices_of_a_wider_tensor(cuda):
    """Depth logits and context as slices of ONE concatenated channels-last tensor (row pitch D + C, the reference's x[:, :D] /
    x[:, D:D+C]): the logits are read in place through their pitch, the mis-aligned context slice is packed by the wrapper."""
    from omnihd_amd import ops
    M, D, C, H, W = 2, 59, 64, 6, 10
    x = torch.randn(M, D + C + 5, H, W, device=cuda).contiguous(memory_format=torch.channels_last)
    depth, rows, feat = ops.depth_head(x[:, :D], x[:, D:D + C], want_rows=True)
    wd, wr, wf = _reference(x[:, :D], x[:, D:D + C])
    assert float((depth - wd).abs().max()) <= 1e-6 and torch.equal(feat.contiguous(), wf)
    # an NCHW (not channels-last) input is packed first: same values
    d2, _, f2 = ops.depth_head(x[:, :D].contiguous(), x[:, D:D + C].contiguous(), want_rows=False)
    assert torch.equal(d2, depth) and torch.equal(f2.contiguous(), feat.contiguous())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_depth_head_backward_matches_autograd_of_the_torch_formulation(cuda, dtype):
    from omnihd_amd import ops
    M, D, C, H, W = 3, 59, 64, 9, 14
    g = torch.Generator(device="cpu").manual_seed(7)
    l0 = (torch.randn(M, D, H, W, generator=g) * 2).to(cuda).to(dtype).contiguous(memory_format=torch.channels_last)
    c0 = torch.randn(M, C, H, W, generator=g).to(cuda).to(dtype).contiguous(memory_format=torch.channels_last)
    w_d, w_r, w_f = (torch.randn(s, generator=g).to(cuda) for s in ((M, D, H, W), (M, H, W, D), (M, H, W, C)))
    grads = []
    for fused in (True, False):
        lg, cx = l0.clone().requires_grad_(), c0.clone().requires_grad_()
        depth, rows, feat = ops.depth_head(lg, cx, want_rows=True) if fused else _reference(lg, cx)
        ((depth * w_d).sum() + (rows * w_r).sum() + (feat * w_f).sum()).backward()
        grads.append((lg.grad.float(), cx.grad.float()))
        assert lg.grad.dtype == dtype and cx.grad.dtype == dtype
    tol = 1e-5 if dtype == torch.float32 else 1e-2            # bf16: one rounding of the result (2^-8) on both sides
    for a, b in zip(grads[0], grads[1]):
        assert float((a - b).abs().max()) <= tol * float(b.abs().max())
    # only one of the two distribution outputs used (the other gradient is absent, not zero-filled)
    lg = l0.clone().requires_grad_()
    depth, rows, feat = ops.depth_head(lg, c0, want_rows=True)
    (rows * w_r).sum().backward()
    lr = l0.clone().requires_grad_()
    (_reference(lr, c0)[1] * w_r).sum().backward()
    assert float((lg.grad.float() - lr.grad.float()).abs().max()) <= tol * float(lr.grad.float().abs().max())


def test_epilogue_plus_pooling_matches_softmax_plus_oracle_pooling_full_size(cuda, golden):
    """DepthNet-shaped logits + context -> depth-head kernel -> the product's pooling launch (kept empty
    rows, twice, so that the second launch really skips them), against torch.softmax + the CPU restatement of the reference
    pooling kernel on the reference-format tables, at the BASELINE frame size R1: every row 1e-5 relative."""
    from omnihd_amd import build_plan, ops
    from omnihd_amd.plan import planned_pool
    geom, dx, bx, nx = full_size_geometry("r1")
    plan = build_plan(t(geom, cuda), dx, bx, nx, layout="bzyx")
    rb, rd, rf, st, ln = O.voxel_pooling_prepare_v2(geom, dx, bx, nx)
    g = torch.Generator(device="cpu").manual_seed(11)
    for rep in range(2):
        logits = (torch.randn(6, 59, 64, 176, generator=g) * 2).contiguous(memory_format=torch.channels_last)
        context = torch.randn(6, 64, 64, 176, generator=g).contiguous(memory_format=torch.channels_last)
        depth, _, feat = ops.depth_head(logits.to(cuda), context.to(cuda))
        got = planned_pool(depth.view(1, 6, 59, 64, 176), feat.view(1, 6, 64, 176, 64), plan, keep_empty_rows=True)
        got = got.permute(0, 2, 3, 4, 1).contiguous().cpu().numpy()
        del depth, feat
        wd = logits.softmax(1).view(1, 6, 59, 64, 176).numpy()
        wf = context.permute(0, 2, 3, 1).contiguous().view(1, 6, 64, 176, 64).numpy()
        want = OC.bev_pool_v2_fwd(np.ascontiguousarray(wd), wf, rd, rf, rb, (1, 16, 160, 240, 64), st, ln, threads=True)
        assert float(np.abs(got - want).max()) <= 1e-5 * float(np.abs(want).max())
        assert np.array_equal((got == 0).all(-1), (want == 0).all(-1))
    assert len(plan._kept_outputs) == 1                      # the second launch reused the first launch's buffer


def test_kept_empty_rows_forward_is_bit_identical_and_never_touches_a_live_result(cuda):
    from omnihd_amd import build_plan
    from omnihd_amd.plan import planned_pool
    geom, dx, bx, nx = full_size_geometry("r1")
    plan = build_plan(t(geom, cuda), dx, bx, nx, layout="byxz")
    g = torch.Generator(device=cuda).manual_seed(3)
    mk = lambda: (torch.rand(1, 6, 59, 64, 176, device=cuda, generator=g), torch.randn(1, 6, 64, 176, 64, device=cuda, generator=g))
    d1, f1 = mk()
    d2, f2 = mk()
    fresh1, fresh2 = planned_pool(d1, f1, plan).clone(), planned_pool(d2, f2, plan).clone()
    a = planned_pool(d1, f1, plan, keep_empty_rows=True)
    assert torch.equal(a, fresh1)
    snapshot = a.clone()
    b = planned_pool(d2, f2, plan, keep_empty_rows=True)         # `a` is alive: a second buffer is used
    assert torch.equal(b, fresh2) and torch.equal(a, snapshot) and a.data_ptr() != b.data_ptr()
    view = a.permute(0, 2, 3, 4, 1)[..., :8]                      # a VIEW keeps the storage in use as well
    del a
    c = planned_pool(d2, f2, plan, keep_empty_rows=True)         # both buffers busy (view of a, b): a plain buffer
    assert torch.equal(c, fresh2) and len(plan._kept_outputs) == 2
    assert torch.equal(view, snapshot.permute(0, 2, 3, 4, 1)[..., :8])
    ptr = view.data_ptr()
    del view, c
    d = planned_pool(d1, f1, plan, keep_empty_rows=True)         # the first buffer is free again: reused, empty rows skipped
    assert d.data_ptr() == ptr and torch.equal(d, fresh1)
    # gradients flow as before
    dd, ff = d1.clone().requires_grad_(), f1.clone().requires_grad_()
    del d
    planned_pool(dd, ff, plan, keep_empty_rows=True).square().sum().backward()
    d0, f0 = d1.clone().requires_grad_(), f1.clone().requires_grad_()
    planned_pool(d0, f0, plan).square().sum().backward()
    assert torch.equal(dd.grad, d0.grad) and torch.equal(ff.grad, f0.grad)


def test_lss_module_uses_the_epilogue_and_matches_the_unfused_path(cuda):
    """LiftSplatShoot_Depth on the tiny rig with and without the fused epilogue (OMNIHD_DEPTH_HEAD=0 = the reference's
    cat / slice / softmax / permute formulation on torch ops): BEV feature, depth distribution, KL depth loss and the
    gradient reaching the image features agree to 1e-5."""
    import os
    from omnihd_amd.harness import TINY, synthetic_lidar2img
    from projects.mmdet3d_plugin.bevfusion.detectors.cam_stream_lss_bevpoolv2_depthnet import LiftSplatShoot_Depth
    tn = TINY
    torch.manual_seed(0)
    net = LiftSplatShoot_Depth(final_dim=(tn["H"], tn["W"]), camera_depth_range=tn["depth_range"], pc_range=tn["pc_range"],
                               downsample=4, grid=tn["grid"], inputC=256, camC=64,
                               norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01)).to(cuda).to(memory_format=torch.channels_last)
    net.eval()                                        # BatchNorm in inference mode: the two runs see the same statistics
    inv = [torch.Tensor(m).inverse() for m in synthetic_lidar2img("tiny")]
    rots = torch.stack([m[:3, :3] for m in inv])[None].to(cuda)
    trans = torch.stack([m[:3, 3] for m in inv])[None].to(cuda)
    x0 = torch.randn(1, 6, 256, tn["H"] // 4, tn["W"] // 4, device=cuda)
    gt = torch.zeros(1, 6, tn["H"], tn["W"], device=cuda)
    gt[:, :, ::3, ::5] = torch.rand(1, 6, (tn["H"] + 2) // 3, (tn["W"] + 4) // 5, device=cuda) * 7 + 1
    res = []
    for flag in ("1", "0"):
        os.environ["OMNIHD_DEPTH_HEAD"] = flag
        try:
            x = x0.clone().requires_grad_()
            bev, depth = net(x, rots, trans)
            assert (getattr(depth, "_omnihd_rows", None) is not None) == (flag == "1")
            loss, _ = net.get_depth_loss(gt, depth, "kld")
            (bev.float().square().mean() + loss).backward()
            res.append((bev.detach().float(), depth.detach().float(), loss.detach(), x.grad.clone()))
        finally:
            os.environ.pop("OMNIHD_DEPTH_HEAD", None)
    for a, b in zip(res[0], res[1]):
        assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-6)
