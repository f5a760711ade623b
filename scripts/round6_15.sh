#!/bin/bash
# Round 6: the fused slab sum of the NHWC weight gradient — parity, A/B in the fp32 step, the one-rank DDP block
export TMPDIR=/tmp; out=gpurun_out/r6_15; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_wgrad_nhwc_gpu.py tests/test_conv_split_gpu.py tests/test_conv_gpu.py tests/test_determinism_gpu.py tests/test_stage_gradients_gpu.py -q > $out/pytest_wgrad.txt 2>&1; echo "pytest rc $?"; tail -5 $out/pytest_wgrad.txt
for f in 0 1 0 1; do
  OMNIHD_WGRAD_FUSED_SUM=$f timeout 300 python3 scripts/lab/fault_repro.py fp32 60 2>/dev/null | tail -1 | sed "s/^/fused_sum=$f fp32: /"
done
OMNIHD_BENCH_DDP1_FRESH=0 timeout 900 python3 bench.py --dtype fp32 --no-cpu-baseline > $out/bench_fp32.json 2> $out/bench_fp32.err; echo "bench rc $?"
python3 - <<PY
import json
l = json.loads(open("$out/bench_fp32.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value", "ms_per_step")}, l["step_ms"]["median"])
d = l.get("ddp_1rank", {})
print("ddp1", {k: d.get(k) for k in ("ms_per_step", "overhead_vs_plain", "overhead_vs_plain_after_median")})
PY
