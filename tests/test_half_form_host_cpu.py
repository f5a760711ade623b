"""Host logic of the TF32-grade half form (round 6, omnihd_amd/ops/planes.py, conv_fp32.py) that needs no GPU: which layers the form
takes, the ring of scale slots (re-zeroed half by half, so that a slot handed to a producer's backward just before a boundary is
still intact when the consumer's cast is enqueued), and the producer -> consumer tags (one half plane / two bf16 planes on the same
attribute: neither consumer may take the other's)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]


def test_which_layers_the_half_form_takes():
    """Stride-1 'same' 1x1 / 3x3 layers with 64-multiple channels in ALL three directions; everything else stays fp32-grade
    (the geometry predicates are host code of the library: omnihd_conv_gen_supported, the NHWC weight-gradient plan)."""
    from omnihd_amd import ops
    w = lambda co, ci, k: torch.empty(co, ci, k, k)
    x = (1, 128, 160, 240)
    assert ops.conv_f16_applies(x, w(256, 128, 3), (1, 1), (1, 1), (1, 1))
    assert ops.conv_f16_applies(x, w(256, 128, 1), (1, 1), (0, 0), (1, 1))
    assert ops.conv_f16_applies((6, 256, 64, 176), w(256, 256, 3), (1, 1), (6, 6), (6, 6))            # ASPP's dilated 3x3
    assert not ops.conv_f16_applies(x, w(256, 128, 3), (2, 2), (1, 1), (1, 1))                          # strided: conv_gen, fp32-grade
    assert not ops.conv_f16_applies(x, w(256, 128, 3), (1, 1), (0, 0), (1, 1))                          # not 'same'
    assert not ops.conv_f16_applies((1, 32, 160, 240), w(64, 32, 3), (1, 1), (1, 1), (1, 1))            # Cin not a multiple of 64
    assert not ops.conv_f16_applies(x, w(72, 128, 3), (1, 1), (1, 1), (1, 1))                           # Cout not a multiple of 64 (data gradient)
    assert not ops.conv_f16_applies(x, w(256, 128, 5), (1, 1), (2, 2), (1, 1))                          # 5x5
    assert not ops.conv_f16_applies(x, w(256, 128, 3).half(), (1, 1), (1, 1), (1, 1))                   # fp32 master weights only


def test_policy_switches(monkeypatch):
    from omnihd_amd import ops
    monkeypatch.delenv("OMNIHD_FP32_CONV", raising=False)
    monkeypatch.delenv("OMNIHD_DETERMINISTIC", raising=False)
    assert ops._fp32_policy() == "tune" and not ops.f16_handover()
    monkeypatch.setenv("OMNIHD_FP32_CONV", "f16")
    assert ops._fp32_policy() == "f16" and ops.f16_handover()
    monkeypatch.setenv("OMNIHD_F16_HANDOVER", "0")
    assert not ops.f16_handover()
    monkeypatch.setenv("OMNIHD_DETERMINISTIC", "1")                      # the deterministic mode pins the fp32-grade split kernels
    assert ops._fp32_policy() == "split"


def test_scale_slots_are_rezeroed_half_by_half(monkeypatch):
    from omnihd_amd import ops
    P = ops.planes
    monkeypatch.setattr(P, "_raw_stream", lambda: 7)
    monkeypatch.setattr(P, "_AMAX_SLOTS", 8)
    monkeypatch.setattr(P, "_AMAX_RING", {})
    dev = torch.device("cpu")
    first = [P._amax_slot(dev) for _ in range(8)]                         # first lap: fresh zeros, nothing re-zeroed
    ring = P._AMAX_RING[(None, 7)][0]
    assert ring.numel() == 16 and all(s.numel() == 2 and s.data_ptr() == ring.data_ptr() + 8 * k for k, s in enumerate(first))
    for s in first:
        s.fill_(3.0)                                                       # every slot used: amax word and inverse scale written
    s0 = P._amax_slot(dev)                                                 # wrap: entering the first half clears IT ...
    assert s0.data_ptr() == ring.data_ptr() and not bool(ring[:8].any()) and bool((ring[8:] == 3.0).all())   # ... and only it
    s0.fill_(5.0)                                                          # a producer's backward accumulates its amax in slot 0
    for _ in range(3):
        P._amax_slot(dev).fill_(5.0)                                       # slots 1-3
    assert bool((ring[:8] == 5.0).all())
    last_of_first_half = ring[6:8]
    s4 = P._amax_slot(dev)                                                 # entering the second half clears the second half only:
    assert s4.data_ptr() == ring.data_ptr() + 8 * 4 and not bool(ring[8:].any())
    assert bool((last_of_first_half == 5.0).all())                         # the slot handed out just before the boundary is intact
    # one ring per (device, stream)
    monkeypatch.setattr(P, "_raw_stream", lambda: 9)
    other = P._amax_slot(dev)
    assert other.data_ptr() != ring.data_ptr() and len(P._AMAX_RING) == 2


def test_half_and_split_tags_do_not_cross(monkeypatch):
    from omnihd_amd import ops
    P = ops.planes
    monkeypatch.setattr(P, "_HALF_WANTED", set())
    monkeypatch.setattr(P, "_PLANES_WANTED", set())
    monkeypatch.setattr(P, "_PLANES_UNUSED", {})
    monkeypatch.setenv("OMNIHD_FP32_CONV", "f16")
    monkeypatch.setenv("OMNIHD_SPLIT_HANDOVER", "1")
    y = torch.randn(2, 8, 4, 4)
    assert P.take_half(y) is None and P.take_planes(y) is None             # untagged: nobody to ask
    P.tag_producer(y, ("bn_y", 1))
    assert P.take_half(y) is None and ("bn_y", 1) in P._HALF_WANTED and P.half_wanted(("bn_y", 1))
    y16 = y.half()
    P.tag_half(y, y16, ("bn_y", 1))
    assert P.take_half(y) is y16
    assert P.take_planes(y) is None and ("bn_y", 1) not in P._PLANES_WANTED    # a split convolution leaves the half plane alone
    hi = y.bfloat16()
    P.tag_planes(y, (hi, (y - hi.float()).bfloat16()), ("bn_y", 1))
    assert P.take_planes(y)[0] is hi
    assert P.take_half(y) is None                                          # ... and the half consumer the bf16 pair (it asks for its own)
    P.tag_half(y, y16, ("bn_y", 1))
    y.add_(1.0)                                                            # written since: the plane is stale
    assert P.take_half(y) is None
    # a producer whose plane nobody takes stops writing it
    for _ in range(9):
        P.tag_half(torch.randn(2, 8, 4, 4), y16, ("bn_y", 2))
    P._HALF_WANTED.add(("bn_y", 3))
    for _ in range(9):
        P.tag_half(torch.randn(2, 8, 4, 4), y16, ("bn_y", 3))
    assert ("bn_y", 3) not in P._HALF_WANTED
    monkeypatch.setenv("OMNIHD_FP32_CONV", "split")
    assert not P.half_wanted(("bn_y", 1))                                  # other policies: no half planes, whoever asked before
    monkeypatch.setenv("OMNIHD_SPLIT_HANDOVER", "0")
    P._PLANES_WANTED.add(("bn_y", 1))
    assert not P.planes_wanted(("bn_y", 1))
    monkeypatch.setenv("OMNIHD_SPLIT_HANDOVER", "all")
    assert P.planes_wanted(("bn_y", 1))
