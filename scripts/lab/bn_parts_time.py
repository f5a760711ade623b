#!/usr/bin/env python3
"""The three device passes of the fp32 BatchNorm backward on one big layer, timed separately (device time behind a spin kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
from omnihd_amd import ops
from omnihd_amd._lib import lib
L = lib()
dev = torch.device("cuda:0")
for shape in ((1, 1024, 160, 240), (1, 512, 160, 240), (6, 256, 64, 176), (1, 64, 160, 240)):
    c = shape[1]; rows = shape[0] * shape[2] * shape[3]
    gy, x, y = (torch.randn(rows, c, device=dev) for _ in range(3))
    gx = torch.empty_like(x)
    local = torch.zeros(2 * c, device=dev); coef = torch.randn(3, c, device=dev)
    ws = torch.empty(L.omnihd_bn_workspace_bytes(rows, c), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(4_000_000); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    p = lambda v: v.data_ptr()
    t_s0 = t(lambda: L.omnihd_bn_channel_sums_f32(p(x), None, None, None, p(local), rows, c, 0, 1.0, p(ws), ws.numel(), st))
    t_s1 = t(lambda: L.omnihd_bn_channel_sums_f32(p(gy), p(x), p(y), None, p(local), rows, c, 1, 1.0, p(ws), ws.numel(), st))
    t_ap = t(lambda: L.omnihd_bn_bwd_apply_f32(p(gy), p(y), None, p(x), p(coef[0]), p(coef[1]), p(coef[2]), p(gx), None, rows, c, st))
    nb = rows * c * 4
    print(f"{shape}: stats(x) {t_s0:6.1f} us = {nb/t_s0/1e6:4.2f} TB/s | sums(gy,x,y) {t_s1:6.1f} us = {3*nb/t_s1/1e6:4.2f} TB/s | apply {t_ap:6.1f} us = {4*nb/t_ap/1e6:4.2f} TB/s")

print("--- other streaming passes of the fp32 step")
for shape in ((1, 1024, 160, 240), (6, 256, 64, 176), (1, 64, 160, 240)):
    c = shape[1]; rows = shape[0] * shape[2] * shape[3]
    x = torch.randn(shape, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(x); scale, shift = torch.randn(c, device=dev), torch.randn(c, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    hi, lo = ops._alloc_planes(x)
    nb = rows * c * 4
    t_split = t(lambda: L.omnihd_split_f32(x.data_ptr(), x.numel(), hi.data_ptr(), lo.data_ptr(), st))
    t_aff = t(lambda: L.omnihd_affine_act_fwd_f32(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, y.data_ptr(), rows, c, 1, st))
    t_affb = t(lambda: L.omnihd_affine_act_bwd_f32(x.data_ptr(), y.data_ptr(), scale.data_ptr(), y.data_ptr(), None, rows, c, 1, st))
    print(f"{shape}: split_f32 {t_split:6.1f} us = {2*nb/t_split/1e6:4.2f} TB/s | affine fwd {t_aff:6.1f} us = {2*nb/t_aff/1e6:4.2f} TB/s | affine bwd {t_affb:6.1f} us = {3*nb/t_affb/1e6:4.2f} TB/s")
