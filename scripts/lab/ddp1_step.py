#!/usr/bin/env python3
"""The R1 fp32 training step, plain or inside a one-rank RCCL group under DistributedDataParallel: ms/step (for rocprofv3 --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
from omnihd_amd.harness import seed_miopen_db
seed_miopen_db()
import torch, torch.distributed as dist
from omnihd_amd.harness import FusionTrainStep
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
torch.cuda.set_device(0)
if mode == "ddp":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29731")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
st = FusionTrainStep(res="r1", batch=1, radar_dims=7, device="cuda:0", seed=1234, dtype="fp32", ddp=mode == "ddp", miopen_find=True)
for _ in range(12):
    st.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 30
for _ in range(n):
    st.step()
torch.cuda.synchronize()
print(mode, "ms/step", round((time.perf_counter() - t0) / n * 1e3, 3))
if mode == "ddp":
    from omnihd_amd import ops
    print(ops.ddp_overlap_info())
    dist.destroy_process_group()
