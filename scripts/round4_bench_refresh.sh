#!/bin/bash
# Round-4 evidence refresh without the PMC passes (bench line, rocprofv3 kernel stats of the same command, fp32 step profile): what
# changes when only host-side code changed (the PMC files stay valid while csrc/bev_pool_v2.hip keeps its hash).
export TMPDIR=/tmp; out=gpurun_out/r4f; mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof -o bench -- python3 bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
python3 - <<PY
import csv, glob
f = glob.glob("$out/prof/**/*kernel_trace.csv", recursive=True)
if f:
    for name in ("k_pool_fwd_direct", "k_pool_bwd_patch", "k_depth_head_fwd", "k_pfn_apply", "k_canvas_nhwc4"):
        rows = [r for r in csv.DictReader(open(f[0])) if name in r["Kernel_Name"]]
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        if d:
            print(name, "launches", len(d), "mean of the last 40 (in-step, second timed run) %.2f us" % (sum(d[-40:]) / len(d[-40:])), "all-launch mean %.2f us" % (sum(d) / len(d)))
            open("$out/%s_durations_us.txt" % name, "w").write("\n".join("%.2f" % v for v in d))
PY
find $out/prof -type f -size +2M -delete
bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_steady.txt 2>&1
find $out -name "*.csv" -size +1M -delete
tail -c 300 $out/bench_default.json; head -4 $out/step_fp32_steady.txt
