"""Host time of a training step, function by function (cProfile over steady steps, GPU left to run behind).  The step is host-bound
once its GPU time falls under the host's enqueue time (the bf16 step since round 5, the TF32-grade fp32 step since round 6):
this is the list of what the host spends it on.
Usage: python scripts/lab/host_profile.py [fp32|bf16] [steps]        (OMNIHD_FP32_CONV etc. from the environment)"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "omnihd-scenes_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from omnihd_amd.harness import FusionTrainStep, seed_miopen_db  # noqa: E402


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    seed_miopen_db()
    st = FusionTrainStep(res="r1", batch=1, radar_dims=7, device="cuda:0", seed=1234, dtype=dt)
    for _ in range(12):
        st.step()
    torch.cuda.synchronize()
    # host-only time of a step: enqueue `steps` steps from an idle GPU and stop the clock BEFORE synchronising
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step()
    t_host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / steps
    print(f"{dt}: enqueue {t_host * 1e3:.2f} ms/step, with the final synchronisation {t_all * 1e3:.2f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        st.step()
    pr.disable()
    torch.cuda.synchronize()
    from omnihd_amd import ops
    print("weight-image table cache:", len(ops._WIMG_TABLES), "entries;", ops.WIMG_STATS)
    for key in ("tottime", "cumtime"):
        print(f"==== by {key} (totals over {steps} steps)")
        pstats.Stats(pr, stream=sys.stdout).strip_dirs().sort_stats(key).print_stats(55)


if __name__ == "__main__":
    main()
