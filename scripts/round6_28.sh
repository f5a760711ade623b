#!/bin/bash
# Round 6: the default bench line on a fourth lease: the heap frozen in front of the warm-up steps
export TMPDIR=/tmp; out=gpurun_out/r6_28; mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
python3 - <<PY
import json
l = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value", "ms_per_step", "step_ms")})
print("tf32_grade", {k: v for k, v in l.get("tf32_grade", {}).items() if k not in ("note", "precision")})
print("per_frame", {k: v for k, v in l["per_frame_calibration"].items() if k in ("ms_per_step", "over_cached_step", "plan_build_us")})
print("roofline", {k: l["roofline"][k] for k in ("mean_launch_us", "frac", "traffic", "bwd_mean_launch_us", "bwd_frac")})
print("bf16", l["bf16_autocast"]["value"], l["bf16_autocast"]["step_ms"])
d = l["ddp_1rank"]; print("ddp1", d["overhead_vs_plain"], d["step_ms"], d.get("overhead_vs_plain_after_median"), d.get("fresh_process", {}).get("overhead_median"))
print("r2", l.get("r2", {}).get("step"))
PY
