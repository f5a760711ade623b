#!/bin/bash
# Round-5 ninth GPU call: tr-read micro-test; deterministic policy (test, pass diagnostic, leftovers, step rate); radar join A/B in the bench.
export TMPDIR=/tmp; out=gpurun_out/r5i; mkdir -p $out
(cd scripts/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tr_read.hip -o tr_read && ./tr_read) > $out/tr_read.txt 2>&1; cat $out/tr_read.txt
timeout 900 python3 -m pytest tests/test_determinism_gpu.py -m gpu -q 2>&1 | tail -5 > $out/det_test.txt; cat $out/det_test.txt
OMNIHD_DETERMINISTIC=1 timeout 900 python3 scripts/lab/determinism_pass.py 2>&1 | grep -v "^/opt\|Warn\|warn" | head -8 > $out/determinism_pass.txt; cat $out/determinism_pass.txt
timeout 600 python3 scripts/lab/det_leftovers.py 2>&1 | grep -v "^/opt\|Warn\|warn" | tail -30 > $out/det_leftovers.txt; cat $out/det_leftovers.txt
OMNIHD_DETERMINISTIC=1 timeout 600 python3 bench.py --dtype fp32 --no-cpu-baseline > $out/bench_deterministic.json 2> $out/bench_deterministic.err; cut -c1-330 $out/bench_deterministic.json
for j in late early late early; do
  OMNIHD_BENCH_DDP1=0 OMNIHD_RADAR_JOIN=$j timeout 600 python3 bench.py --dtype fp32 --no-cpu-baseline > $out/bench_join_$j.json 2> /dev/null
  python3 - <<PY
import json
l = json.loads(open("$out/bench_join_$j.json").readline()); r = l["roofline"]
print("join=$j", "ms/step", l["ms_per_step"], l["step_ms"], "pool fwd in step", r["mean_launch_us"], "frac", r["frac"], "bwd", r["bwd_mean_launch_us"])
PY
done
