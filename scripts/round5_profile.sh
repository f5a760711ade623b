#!/bin/bash
# Round-5 evidence run on the GPU box (one MI355X).  Outputs under gpurun_out/r5p_final/; the summaries are copied into profiles/round5/.
#   1. the default bench command (JSON line incl. cpu_baseline)
#   2. rocprofv3 --kernel-trace --stats of the same command (kernel stats csv + every pooling launch's duration)
#   3. steady-state step profiles (fp32 / bf16): hand-written vs library GPU time, launches per step
#   4. SQ counters of the row-shift convolution kernel (split and bf16 forms) and of the NHWC weight gradient on 1024->1024 @160x240
#   5. the fp32 bench line under OMNIHD_DETERMINISTIC=1
export TMPDIR=/tmp; out=gpurun_out/r5p_final; mkdir -p $out
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
OMNIHD_BENCH_CHILD=1 rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof -o bench -- python3 bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
python3 - <<PY
import csv, glob
f = glob.glob("$out/prof/**/*kernel_trace.csv", recursive=True)
if f:
    for name in ("k_pool_fwd_direct", "k_pool_bwd_patch"):
        rows = [r for r in csv.DictReader(open(f[0])) if name in r["Kernel_Name"]]
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        if d:
            print(name, "launches", len(d), "mean of the last 40 (in-step, second timed run) %.2f us" % (sum(d[-40:]) / len(d[-40:])), "all-launch mean %.2f us" % (sum(d) / len(d)))
            open("$out/%s_durations_us.txt" % name, "w").write("\n".join("%.2f" % v for v in d))
PY
find $out/prof -type f -size +2M -delete
STEP_PROFILE_OUT=$out/fp32 bash scripts/lab/step_profile.sh fp32 6 > $out/step_fp32_steady.txt 2>&1
STEP_PROFILE_OUT=$out/bf16 bash scripts/lab/step_profile.sh bf16 8 > $out/step_bf16_steady.txt 2>&1
bash scripts/lab/pmc_conv.sh $out/pmc_conv_split 1 160 240 1024 1024 3 0 split > $out/conv_rs_split_pmc.txt 2>&1
bash scripts/lab/pmc_conv.sh $out/pmc_conv_bf16 1 160 240 1024 1024 3 0 > $out/conv_rs_bf16_pmc.txt 2>&1
bash scripts/lab/pmc_wgrad.sh $out/pmc_wgrad 1,160,240,1024,1024,3,1,1 > $out/wgrad_nhwc_pmc_final.txt 2>&1
OMNIHD_DETERMINISTIC=1 OMNIHD_BENCH_DDP1=0 python3 bench.py --dtype fp32 > $out/bench_deterministic.json 2> $out/bench_deterministic.err
find $out -name "*.csv" -size +1M -delete
tail -c 600 $out/bench_default.json; head -4 $out/step_fp32_steady.txt; head -4 $out/step_bf16_steady.txt; head -12 $out/conv_rs_split_pmc.txt; cut -c1-300 $out/bench_deterministic.json
