#!/bin/bash
# Round 6: this round's PMC passes of the pooling kernels (R1, R2) + unit counters, then the default bench line and its rocprofv3 kernel stats
export TMPDIR=/tmp; out=gpurun_out/r6_14; mkdir -p $out
bash scripts/lab/pmc_bwd.sh $out/pmc_r1 r1 > $out/pmc_r1.log 2>&1; cp $out/pmc_r1/pmc_pool_r1.json $out/ 2>/dev/null
bash scripts/lab/pmc_bwd.sh $out/pmc_r2 r2 > $out/pmc_r2.log 2>&1; cp $out/pmc_r2/pmc_pool_r2.json $out/ 2>/dev/null
bash scripts/lab/pmc_pool_units.sh $out/units_r1 r1 > $out/units_r1.log 2>&1; cp $out/units_r1/pmc_units_r1.json $out/ 2>/dev/null
bash scripts/lab/pmc_pool_units.sh $out/units_r2 r2 > $out/units_r2.log 2>&1; cp $out/units_r2/pmc_units_r2.json $out/ 2>/dev/null
mkdir -p profiles/round6; cp $out/pmc_pool_r1.json $out/pmc_pool_r2.json profiles/round6/ 2>/dev/null     # so that the bench below finds this round's traffic
python3 - <<PY
import json
for r in ("r1", "r2"):
    try:
        d = json.load(open("$out/pmc_pool_%s.json" % r))
        for k in ("fwd_lean", "patch_bwd"):
            print(r, k, {x: round(d[k][x]) if d[k][x] > 10 else round(d[k][x], 4) for x in ("read_bytes_corrected", "write_bytes", "l2_hit_rate", "l1_miss_share") if x in d[k]})
    except Exception as e:
        print(r, "no pmc", e)
PY
timeout 1500 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
tail -2 $out/bench_default.err | cut -c1-200
python3 - <<PY
import json
l = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value", "ms_per_step", "step_ms", "dtype")})
print("bf16", {k: v for k, v in l.get("bf16_autocast", {}).items() if k != "note"})
print("per_frame", {k: v for k, v in l["per_frame_calibration"].items() if k != "note"})
print("plan_build", {k: v for k, v in l["ops_roofline"]["plan_build"].items() if k != "note"})
print("roofline", {k: l["roofline"][k] for k in ("mean_launch_us", "frac", "traffic", "bwd_mean_launch_us", "bwd_frac", "frac_vs_copy_peak", "bwd_in_step_over_isolated")})
print("r2", {k: l["r2"][k] for k in ("fwd_warm_us", "fwd_frac", "bwd_warm_us", "bwd_frac", "fwd_traffic")}, l["r2"].get("step", {}).get("ms_per_step"))
print("ddp1", {k: v for k, v in l.get("ddp_1rank", {}).items() if k in ("ms_per_step", "overhead_vs_plain")}, l.get("ddp_1rank", {}).get("fresh_process", {}))
print("cpu", l.get("cpu_baseline", {}).get("value"), l.get("fp32_library"))
PY
find $out -name "*.csv" -size +1M -delete
