#!/usr/bin/env python3
"""What makes the pooling forward slower inside the training step (62 us) than back to back (44 us)?  Time ONE pool launch
(events around it) right after (a) another pool launch, (b) a 1 GB device copy, (c) a burst of bf16 GEMMs, (d) a long idle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import bench
wl = bench.BevOps("r1", 1, torch.device("cuda:0"), 1234)
big_a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda")   # 1 GiB
big_b = torch.empty_like(big_a)
ga = torch.randn(8192, 8192, device="cuda").bfloat16()
gb = torch.randn(8192, 8192, device="cuda").bfloat16()


def timed_pool(pre, n=12):
    out = []
    for k in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        wl.pool_fwd(k % 4)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return sorted(out)[len(out) // 2], min(out)


for k in range(300):
    wl.pool_fwd(k % 4)
torch.cuda.synchronize()
print("median / min us of one pool launch right after ...")
print("  another pool launch      ", timed_pool(lambda: wl.pool_fwd(3)))
print("  1 GiB device copy        ", timed_pool(lambda: big_b.copy_(big_a)))
print("  4 bf16 GEMMs 8192^3      ", timed_pool(lambda: [torch.mm(ga, gb) for _ in range(4)]))
print("  40 bf16 GEMMs 8192^3     ", timed_pool(lambda: [torch.mm(ga, gb) for _ in range(40)], n=6))
print("  30 ms host sleep (idle)  ", timed_pool(lambda: (torch.cuda.synchronize(), time.sleep(0.03))))
print("  copy then 4 GEMMs        ", timed_pool(lambda: (big_b.copy_(big_a), [torch.mm(ga, gb) for _ in range(4)])))


def touch(stride_bytes):
    def f(s):
        depth, feat, og, out, dg, fg, tb = wl.sets[s]
        acc = 0
        for t in (out, depth, feat, tb[0], tb[2], tb[8]):
            v = t.view(-1)
            step = max(1, stride_bytes // v.element_size())
            acc = acc + v[::step].float().sum()
        return acc
    return f


def timed_pool_touch(pre, stride, n=12):
    """[touch + pool] timed together after `pre`"""
    out = []
    tf = touch(stride)
    for k in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tf(k % 4)
        wl.pool_fwd(k % 4)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return sorted(out)[len(out) // 2], min(out)


print("after a 1 GiB copy, [touch every 2 MiB + pool]   ", timed_pool_touch(lambda: big_b.copy_(big_a), 2 << 20))
print("after a 1 GiB copy, [touch every 64 KiB + pool]  ", timed_pool_touch(lambda: big_b.copy_(big_a), 64 << 10))
print("after a 1 GiB copy, [touch every 4 KiB + pool]   ", timed_pool_touch(lambda: big_b.copy_(big_a), 4 << 10))
print("after a pool launch, [touch every 2 MiB + pool]  ", timed_pool_touch(lambda: wl.pool_fwd(3), 2 << 20))

tiny = torch.zeros(64, device="cuda")


def pool_only(pre, n=12):
    out = []
    for k in range(n):
        pre(k % 4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        wl.pool_fwd(k % 4)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return sorted(out)[len(out) // 2], min(out)


def unrelated(s):
    big_b.copy_(big_a)
    for _ in range(18):
        tiny.add_(1.0)


def touched(stride):
    tf = touch(stride)

    def f(s):
        big_b.copy_(big_a)
        tf(s)
    return f


def touched_out_only(s):
    big_b.copy_(big_a)
    wl.sets[s][3].view(-1)[::524288].float().sum()


print("pool alone after copy + 18 tiny unrelated kernels ", pool_only(unrelated))
print("pool alone after copy + touch every 2 MiB (6 bufs)", pool_only(touched(2 << 20)))
print("pool alone after copy + touch of `out` only       ", pool_only(touched_out_only))

big_c = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda")   # 512 MiB, read-only sweeps


def copy_then_read_sweep(s):
    big_b.copy_(big_a)
    big_c.sum()                      # 512 MiB of reads: evicts (writes back) the copy's dirty lines before the pool starts


def read_sweep_only(s):
    big_c.sum()


print("pool alone after copy + 512 MiB read sweep        ", pool_only(copy_then_read_sweep))
print("pool alone after 512 MiB read sweep only          ", pool_only(read_sweep_only))

from omnihd_amd import ops
small_desc = [wl.sets[s][6][8][:64].clone() for s in range(4)]     # 64 descriptors: 8 workgroups per XCD


def sweep_then_tiny_pool(s):
    big_c.sum()
    depth, feat, og, out, dg, fg, tb = wl.sets[(s + 1) % 4]
    ops.bev_pool_v2_forward_lean(depth, feat, tb[0], tb[2], small_desc[(s + 1) % 4], out, wl.D, wl.fH * wl.fW)


print("pool alone after read sweep + a 64-tile pool launch on ANOTHER buffer set (code warm, data cold)", pool_only(sweep_then_tiny_pool))


def series(n_sets):
    big_c.sum()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    ev[0].record()
    for k in range(8):
        wl.pool_fwd(k % n_sets)
        ev[k + 1].record()
    torch.cuda.synchronize()
    return [round(ev[k].elapsed_time(ev[k + 1]) * 1e3, 1) for k in range(8)]


for _ in range(2):
    print("8 launches after a read sweep, rotating 4 sets:", series(4))
for _ in range(2):
    print("8 launches after a read sweep, always set 0   :", series(1))
for _ in range(2):
    print("8 launches after a read sweep, rotating 2 sets:", series(2))


def sweep_then_prefetch(which):
    def f(s):
        big_c.sum()
        depth, feat, og, out, dg, fg, tb = wl.sets[s]
        bufs = {"tables": [tb[8], tb[2], tb[0]], "tables+depth+feat": [tb[2], tb[0], depth, feat], "depth+feat": [depth, feat]}[which]
        ops.prefetch(bufs)
        torch.cuda.current_stream().wait_stream(ops._PREFETCH_STREAMS[0])
    return f


for which in ("tables", "depth+feat", "tables+depth+feat"):
    print(f"pool alone after read sweep + read-ahead of {which:18s}", pool_only(sweep_then_prefetch(which)))

print("lean2 with in-kernel read-ahead (this build): pool alone after read sweep                 ", pool_only(read_sweep_only))
print("lean2 with in-kernel read-ahead (this build): pool alone after sweep + tables read-ahead  ", pool_only(sweep_then_prefetch("tables")))
print("lean2 with in-kernel read-ahead (this build): back to back                                ", pool_only(lambda s: wl.pool_fwd(3)))
