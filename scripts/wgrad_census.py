#!/usr/bin/env python3
"""Every distinct convolution geometry that reaches the MFMA weight-gradient kernel in one training step,
with its count, and the time of (a) our kernel chain and (b) MIOpen's weight gradient for the same shapes."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops
from omnihd_amd.harness import FusionTrainStep

seen = collections.Counter()
real = ops.conv_wgrad


def spy(x, g, k, stride=1, padding=0, dilation=1):
    seen[(tuple(x.shape), tuple(g.shape), int(k), int(stride), int(padding), int(dilation))] += 1
    return real(x, g, k, stride, padding, dilation)


ops.conv_wgrad = spy
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, dtype="bf16", miopen_find=True)
st.step(); st.step()
seen.clear()
st.step()
torch.cuda.synchronize()
ops.conv_wgrad = real


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot_ours = tot_mi = tot_best = 0.0
rows = []
for (xs, gs, k, s, p, d), cnt in sorted(seen.items(), key=lambda kv: -kv[1]):
    x = torch.randn(xs, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    g = torch.randn(gs, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(gs[1], xs[1], k, k, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    t_ours = bench(lambda: ops.conv_wgrad(x, g, k, s, p, d))
    t_mi = bench(lambda: torch.ops.aten.convolution_backward(g, x, w, None, [s, s], [p, p], [d, d], False, [0, 0], 1,
                                                             [False, True, False]))
    flops = 2.0 * gs[0] * gs[2] * gs[3] * gs[1] * xs[1] * k * k
    rows.append((cnt, xs, gs[1], k, s, p, d, t_ours, t_mi, flops))
    tot_ours += cnt * t_ours; tot_mi += cnt * t_mi; tot_best += cnt * min(t_ours, t_mi)
print(f"{'n':>3} {'x shape':>22} {'cout':>5} k s  p  d   ours us   miopen us  GFLOP  ratio")
for cnt, xs, co, k, s, p, d, a, b, fl in rows:
    print(f"{cnt:3d} {str(xs):>22} {co:5d} {k} {s} {p:2d} {d:2d} {a:9.1f} {b:10.1f} {fl/1e9:7.2f} {b/a:6.2f}")
print(f"per step: ours {tot_ours/1e3:.2f} ms, miopen {tot_mi/1e3:.2f} ms, best-of {tot_best/1e3:.2f} ms over {sum(seen.values())} convs")
