#!/bin/bash
# Round-4 evidence run on the GPU box (one MI355X).  Outputs under gpurun_out/r4p/; the summaries are copied into profiles/round4/.
#   1. the default bench command (JSON line incl. cpu_baseline)
#   2. rocprofv3 --kernel-trace --stats of the same command (kernel stats csv + every pooling launch's duration)
#   3. PMC traffic of the pooling kernels (separate --pmc passes, FETCH_SIZE calibrated on a copy)
#   4. steady-state step profiles (fp32 / bf16): hand-written vs library GPU time, launches per step
#   5. SQ counters of the row-shift convolution kernel (split and bf16 forms) on 1024->1024 @160x240
#   6. convolution kernels against MIOpen per geometry (bf16 and fp32-grade split)
export TMPDIR=/tmp; out=gpurun_out/r4p; mkdir -p $out
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
bash scripts/lab/pmc_bwd.sh $out/pmc_r1 r1 > $out/pmc_r1.log 2>&1
bash scripts/lab/pmc_bwd.sh $out/pmc_r2 r2 > $out/pmc_r2.log 2>&1
bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_steady.txt 2>&1
bash scripts/lab/step_profile.sh bf16 8 > $out/step_bf16_steady.txt 2>&1
find $out -name "*.csv" -size +1M -delete
tail -c 1500 $out/bench_default.json; tail -3 $out/pmc_r1.log; head -4 $out/step_fp32_steady.txt; head -4 $out/step_bf16_steady.txt
