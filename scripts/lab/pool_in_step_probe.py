#!/usr/bin/env python3
"""Why is the pooling forward slower inside the training step than after a rewrite of its inputs (scripts/lab/mall_probe.py)?

Runs the fusion training step and times the pooling launches inside it (events of omnihd_amd.plan.TIMING) with a pre-action
inserted right in front of the forward launch:
  none       the step as it is
  readahead  depth + feat (the tensors handed to the kernel) read ahead on the side stream, waited for
  tables     the three plan tables read ahead, waited for (the product reads them ahead un-waited)
  both
  idle       a device-wide synchronisation before the launch (queue empty, clocks settle)
Usage: pool_in_step_probe.py [bf16|fp32] [steps]   (OMNIHD_POOL_READAHEAD=0 to switch the in-kernel read-ahead off)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: E402,F401  (seeds MIOpen's database)
import torch  # noqa: E402

import omnihd_amd.plan as plan_mod  # noqa: E402
from omnihd_amd import ops  # noqa: E402
from omnihd_amd.harness import FusionTrainStep  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = FusionTrainStep(res="r1", batch=1, radar_dims=7, device="cuda:0", seed=1234, dtype=dt, miopen_find=True)
for _ in range(4):
    wl.step()
torch.cuda.synchronize()

MODE = ["none"]
orig_lean = ops.bev_pool_v2_forward_lean
orig_timed = plan_mod._timed


def timed(kind, launch):
    if kind == "fwd" and MODE[0] != "none":
        args = launch.__closure__
        cells = {type(c.cell_contents).__name__: c.cell_contents for c in args}
        depth = feat = plan = None
        for c in args:
            v = c.cell_contents
            if isinstance(v, torch.Tensor) and v.dim() == 5 and v.dtype == torch.float32:
                if v.shape[-1] == 64 and feat is None and v.shape[2] != 59:
                    feat = v
                else:
                    depth = v
            elif hasattr(v, "tile_desc"):
                plan = v
        bufs = []
        if MODE[0] in ("readahead", "both"):
            bufs += [depth, feat]
        if MODE[0] in ("tables", "both"):
            ops.prefetch([plan.tile_desc, plan.row_ptr, plan.ranks_depth])
        if bufs:
            ops.prefetch(bufs)
        if MODE[0] == "idle":
            torch.cuda.synchronize()
        elif MODE[0] != "none":
            torch.cuda.current_stream().wait_stream(ops._PREFETCH_STREAMS[0])
    return orig_timed(kind, launch)


plan_mod._timed = timed
print("dtype", dt, "| in-kernel read-ahead:", os.environ.get("OMNIHD_POOL_READAHEAD", "1"))
for mode in ("none", "readahead", "tables", "both", "idle", "none"):
    MODE[0] = mode
    plan_mod.TIMING = []
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    t = {}
    for kind, e0, e1 in plan_mod.TIMING:
        t.setdefault(kind, []).append(e0.elapsed_time(e1) * 1e3)
    plan_mod.TIMING = None
    print("%-10s" % mode, {k: "mean %.1f min %.1f max %.1f us (%d)" % (sum(v) / len(v), min(v), max(v), len(v)) for k, v in t.items()})
