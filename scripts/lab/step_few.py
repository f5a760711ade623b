#!/usr/bin/env python3
"""A few R1 training steps (kernel-trace workload): python3 scripts/lab/step_few.py [bf16|fp32] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import bench  # noqa: F401  (seeds the MIOpen user db)
import torch
from omnihd_amd.harness import FusionTrainStep
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ddp = os.environ.get("OMNIHD_STEP_DDP", "0") == "1"       # the same step inside a one-rank RCCL group under DistributedDataParallel
if ddp:
    import torch.distributed as dist
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29733")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
st = FusionTrainStep(res=os.environ.get("OMNIHD_STEP_RES", "r1"), batch=1, radar_dims=7, dtype=dt, ddp=ddp,
                     miopen_find=os.environ.get("OMNIHD_STEP_FIND", "1") == "1")
for _ in range(3 if not ddp else 5):
    st.step()
torch.cuda.synchronize()
print("MARK timed steps begin", flush=True)
for _ in range(n):
    st.step()
torch.cuda.synchronize()
print("done")
if ddp:
    from omnihd_amd import ops
    print(ops.ddp_overlap_info(), ops.fast_paths_report()["wgrad_overlap"])
    dist.destroy_process_group()
