"""The cached kernel images of master weights (bf16 shadows, split planes) must notice EVERY optimiser step.
Round 3 found that torch.optim.AdamW(fused=True) updates parameters without moving their autograd version counter, which the
caches were keyed on: after the first step the convolutions kept reading step-0 weights.  The caches are now also keyed on a
generation counter moved by torch's global optimiser-step hook (omnihd_amd/ops/weights.py::_WEIGHT_GEN)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]


@pytest.mark.parametrize("kw", [dict(fused=True), dict(foreach=True), dict()])
def test_bf16_image_follows_the_optimiser(kw):
    from omnihd_amd import ops
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(16, 8, 3, 3))
    try:
        opt = torch.optim.AdamW([w], lr=0.1, **kw)
    except (RuntimeError, TypeError):
        pytest.skip("this optimiser variant does not exist for CPU tensors here")
    first = ops.bf16_of(w).clone()
    assert torch.equal(first, w.detach().to(torch.bfloat16))
    for _ in range(2):
        w.grad = torch.randn_like(w)
        opt.step()
        img = ops.bf16_of(w)
        assert torch.equal(img, w.detach().to(torch.bfloat16)), "stale bf16 image after an optimiser step"
    assert not torch.equal(first, ops.bf16_of(w))


def test_generation_counter_moves_with_every_optimiser_step_and_on_request():
    from omnihd_amd import ops
    w = torch.nn.Parameter(torch.zeros(4))
    g0 = ops._WEIGHT_GEN[0]
    opt = torch.optim.SGD([w], lr=0.1)
    w.grad = torch.ones(4)
    opt.step()
    assert ops._WEIGHT_GEN[0] == g0 + 1
    ops.weights_changed()
    assert ops._WEIGHT_GEN[0] == g0 + 2
    v = ops._wver(w)
    with torch.no_grad():
        w.add_(1.0)                       # writes through the dispatcher move the version counter
    assert ops._wver(w) != v


def test_writes_through_data_need_the_explicit_notice():
    """``param.data.add_()`` moves neither the version counter nor an optimiser step: the documented contract is to call
    ops.weights_changed() afterwards."""
    from omnihd_amd import ops
    w = torch.nn.Parameter(torch.randn(8, 8, 1, 1))
    a = ops.bf16_of(w).clone()
    w.data.add_(1.0)
    ops.weights_changed()
    b = ops.bf16_of(w)
    assert torch.equal(b, w.detach().to(torch.bfloat16)) and not torch.equal(a, b)


def test_one_launch_refresh_follows_a_parameter_that_gets_new_storage():
    """``refresh_bf16_shadows`` keeps a launch plan with the master weights' raw pointers / detached aliases.  A
    Parameter that keeps its identity but gets NEW storage (``param.data = ...``, ``module.to(memory_format=...)``) moves neither
    its version counter nor the set of registered images: the plan must notice (pointer + strides are re-checked every call)
    and copy from the live tensor, not from the freed one."""
    from omnihd_amd import ops
    torch.manual_seed(1)
    w = torch.nn.Parameter(torch.randn(16, 8, 3, 3))
    v = torch.nn.Parameter(torch.randn(32))
    ops.bf16_of(w); ops.bf16_of(v)
    assert ops.refresh_bf16_shadows() >= 2
    old_w, old_v = w.data, v.data                             # keep the old storages alive: a stale plan would copy from them
    w.data = torch.randn(16, 8, 3, 3).contiguous(memory_format=torch.channels_last)       # new storage AND new strides
    v.data = torch.randn(32)
    ops.weights_changed()                                     # the documented notice for writes outside torch's optimisers
    ops.refresh_bf16_shadows()
    assert torch.equal(ops.bf16_of(w), w.detach().to(torch.bfloat16)), "bf16 image copied from the parameter's OLD storage"
    assert torch.equal(ops.bf16_of(v), v.detach().to(torch.bfloat16))
    assert not torch.equal(ops.bf16_of(w), old_w.to(torch.bfloat16)) and not torch.equal(ops.bf16_of(v), old_v.to(torch.bfloat16))
