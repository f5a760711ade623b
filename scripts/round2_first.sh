#!/bin/bash
# First GPU call of round 2: everything that was written after round 1's GPU budget ended, in order of importance.
#   gpurun --timeout 1500 -- 'bash scripts/round2_first.sh'
# Logs under gpurun_out/r2_first/.  Each leg has its own time limit so that one slow leg cannot eat the call.
set -u
out=gpurun_out/r2_first
mkdir -p "$out"
export TMPDIR=/tmp

# 1. the regular GPU suite (geometry is now broadcast multiply-adds everywhere: first full run on the device)
timeout 700 python -m pytest tests -m gpu -x -q > "$out/gpu_suite.log" 2>&1; echo "gpu suite rc=$?" | tee -a "$out/summary.txt"
# 2. pending checks: packed HardVFE at LiDAR sizes
OMNIHD_TEST_PENDING=1 timeout 200 python -m pytest tests/test_pillars_gpu.py -m gpu -x -q -k packed > "$out/pending.log" 2>&1
echo "pending rc=$?" | tee -a "$out/summary.txt"
# 3. bench (step time with the exact in-step rank tables; DESIGN section 5 caveat)
timeout 420 python bench.py --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" | tee -a "$out/summary.txt"
# 4. BASELINE configs[4]: first measurement, dense and packed voxel encoder
timeout 400 python scripts/try_triple.py 4 2 4 > "$out/triple_dense.log" 2>&1; echo "triple dense rc=$?" | tee -a "$out/summary.txt"
OMNIHD_VFE_PACKED=1 timeout 300 python scripts/try_triple.py 4 2 4 > "$out/triple_packed.log" 2>&1
echo "triple packed rc=$?" | tee -a "$out/summary.txt"
# 5. inference frames/s with the reference tool's protocol
timeout 240 python scripts/infer_fps.py r1 bf16 200 > "$out/infer_fps.log" 2>&1; echo "infer rc=$?" | tee -a "$out/summary.txt"
tail -n 3 "$out"/*.log "$out/bench.json" 2>/dev/null | cut -c1-400
