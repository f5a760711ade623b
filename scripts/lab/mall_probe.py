#!/usr/bin/env python3
"""Do device WRITES leave their lines in the Infinity Cache for the next kernel's gathers?

The pooling forward takes 45 us with its inputs resident in the 256 MiB Infinity Cache and 62-66 us inside the training step,
where the kernels right in front of it WROTE depth / feat (scripts/lab/pool_context.py).  This probe times one pooling launch
(in-kernel read-ahead off: OMNIHD_POOL_READAHEAD=0) after a 512 MiB read sweep followed by
  a) nothing                        (everything cold)
  b) a read-ahead of the plan tables
  c) b + depth / feat REWRITTEN by plain device stores (torch copy kernels from other buffers)
  d) b + depth / feat rewritten IN PLACE (read-modify-write: x.mul_(1))
  e) b + a read-ahead of depth / feat (the known-good case)
If (c) is as slow as (b), stores do not allocate in the Infinity Cache and a producer kernel cannot leave its output warm."""
import os
import sys

os.environ.setdefault("OMNIHD_POOL_READAHEAD", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from omnihd_amd import ops  # noqa: E402

wl = bench.BevOps("r1", 1, torch.device("cuda:0"), 1234)
big_c = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda")
src = [(s[0].clone(), s[1].clone()) for s in wl.sets]


def pool_only(pre, n=16):
    out = []
    for k in range(n):
        big_c.sum()
        pre(k % 4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        wl.pool_fwd(k % 4)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    out.sort()
    return "median %.1f  min %.1f us" % (out[len(out) // 2], out[0])


def tables(s):
    tb = wl.sets[s][6]
    ops.prefetch([tb[8], tb[2], tb[0]])
    torch.cuda.current_stream().wait_stream(ops._PREFETCH_STREAMS[0])


def rewrite(s):
    wl.sets[s][0].copy_(src[s][0])
    wl.sets[s][1].copy_(src[s][1])
    tables(s)


def rmw(s):
    wl.sets[s][0].mul_(1.0)
    wl.sets[s][1].mul_(1.0)
    tables(s)


def readahead(s):
    depth, feat, og, out, dg, fg, tb = wl.sets[s]
    ops.prefetch([tb[2], tb[0], depth, feat])
    ops.prefetch([tb[8]])
    torch.cuda.current_stream().wait_stream(ops._PREFETCH_STREAMS[0])


for k in range(300):
    wl.pool_fwd(k % 4)
torch.cuda.synchronize()
print("in-kernel read-ahead:", os.environ["OMNIHD_POOL_READAHEAD"])
for name, fn in (("a) sweep only", lambda s: None), ("b) + tables read ahead", tables), ("c) + depth/feat rewritten (copy)", rewrite),
                 ("d) + depth/feat read-modify-write", rmw), ("e) + depth/feat read ahead", readahead)):
    print("%-40s %s" % (name, pool_only(fn)))
    print("%-40s %s" % (name + " (again)", pool_only(fn)))
