#!/bin/bash
# Round 6 final evidence: default bench line (now with tf32_grade), rocprofv3 kernel stats of the same command, steady-state step profiles
export TMPDIR=/tmp; out=gpurun_out/r6_25; mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?"
OMNIHD_BENCH_CHILD=1 rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof -o bench -- python3 bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err; echo "prof rc $?"
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
python3 - <<PY
import csv, glob
f = glob.glob("$out/prof/**/*kernel_trace.csv", recursive=True)
if f:
    for name in ("k_pool_fwd_direct", "k_pool_bwd_patch", "k_plan_keys"):
        rows = [r for r in csv.DictReader(open(f[0])) if name in r["Kernel_Name"]]
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        if d:
            print(name, "launches", len(d), "mean of the last 40 %.2f us" % (sum(d[-40:]) / len(d[-40:])), "all-launch mean %.2f us" % (sum(d) / len(d)))
            open("$out/%s_durations_us.txt" % name, "w").write("\n".join("%.2f" % v for v in d))
PY
find $out/prof -type f -size +2M -delete
STEP_PROFILE_OUT=$out/fp32 bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_steady.txt 2>&1
STEP_PROFILE_OUT=$out/bf16 bash scripts/lab/step_profile.sh bf16 8 > $out/step_bf16_steady.txt 2>&1
find $out -name "*.csv" -size +1M -delete
python3 - <<PY
import json
l = json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value", "ms_per_step", "step_ms")})
print("tf32_grade", {k: v for k, v in l.get("tf32_grade", {}).items() if k not in ("note", "precision")})
print("per_frame", {k: v for k, v in l["per_frame_calibration"].items() if k != "note"})
print("roofline", {k: l["roofline"][k] for k in ("mean_launch_us", "frac", "traffic", "bwd_mean_launch_us", "bwd_frac")})
print("bf16", l["bf16_autocast"]["value"], l["bf16_autocast"]["step_ms"]); print("ddp1", l["ddp_1rank"]["overhead_vs_plain"], l["ddp_1rank"].get("fresh_process", {}).get("overhead_median"))
print("fast", l.get("fast_paths"))
PY
head -4 $out/step_fp32_steady.txt; head -4 $out/step_bf16_steady.txt
