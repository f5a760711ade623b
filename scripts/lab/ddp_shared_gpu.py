#!/usr/bin/env python3
"""Two ranks SHARING cuda:0 over gloo through the HIP path of the training step (DDP bucket all-reduce + naiveSyncBN / fused-PFN
statistic exchanges + dual-stream forward): a multi-rank run of the GPU code path on a one-GPU box (RCCL refuses two ranks on one
device).  usage: ddp_shared_gpu.py [tiny|r1] [steps]   — prints per-rank losses, step times and whether the replicas stayed identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "omnihd-scenes_amd")]
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, res, steps, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from omnihd_amd.harness import FusionTrainStep, comm_report
    torch.cuda.set_device(0)
    t0 = time.time()
    st = FusionTrainStep(res=res, batch=1, radar_dims=7, device="cuda:0", seed=100 + rank, dtype="fp32", ddp=True, sets=1)
    print(f"rank {rank}: model ready after {time.time() - t0:.1f} s", flush=True)
    losses, times = [], []
    for i in range(steps):
        t1 = time.time()
        losses.append(float(st.step().detach()))
        torch.cuda.synchronize()
        times.append(time.time() - t1)
        print(f"rank {rank}: step {i} loss {losses[-1]:.5f} in {times[-1]:.2f} s", flush=True)
    flat = torch.cat([p.detach().reshape(-1) for p in st.raw_model.parameters()]).cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    comm = comm_report(st, iters=1)
    out[rank] = (losses, times, bool(torch.equal(gathered[0], gathered[1])), comm)
    dist.destroy_process_group()


if __name__ == "__main__":
    res = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with mp.Manager() as man:
        out = man.dict()
        mp.spawn(worker, args=(2, port, res, steps, out), nprocs=2, join=True)
        r = dict(out)
    for k in (0, 1):
        print("rank", k, "losses", r[k][0], "step s", ["%.2f" % t for t in r[k][1]], "replicas identical", r[k][2], "comm", r[k][3])
    assert r[0][2] and r[1][2] and r[0][0] != r[1][0]
    print("OK")
