#!/bin/bash
# Round-2 evidence run on the GPU box: rocprofv3 kernel stats of the bench command, PMC traffic of the pooling kernels,
# MIOpen user db capture (to seed the next fresh box).  Outputs under gpurun_out/r2p/.
export TMPDIR=/tmp; out=gpurun_out/r2p; mkdir -p $out
export MIOPEN_USER_DB_PATH=$PWD/$out/miopen_db MIOPEN_CUSTOM_CACHE_DIR=$PWD/$out/miopen_db
mkdir -p $MIOPEN_USER_DB_PATH; cp omnihd-scenes_amd/miopen_db/* $MIOPEN_USER_DB_PATH/ 2>/dev/null
rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof -o bench -- python3 bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
python3 - <<PY
import csv, glob
f = glob.glob("$out/prof/**/*kernel_trace.csv", recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "k_pool_fwd_lean2" in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    print("pool fwd launches", len(d), "first 8 (us)", [round(v, 1) for v in d[:8]], "last 40 mean", sum(d[-40:]) / max(len(d[-40:]), 1))
    open("$out/pool_fwd_durations_us.txt", "w").write("\n".join("%.2f" % v for v in d))
PY
find $out/prof -type f -size +2M -delete
unset MIOPEN_USER_DB_PATH MIOPEN_CUSTOM_CACHE_DIR
bash scripts/lab/pmc_bwd.sh $out/pmc r1 > $out/pmc.log 2>&1
du -sh $out/miopen_db; ls -la $out/miopen_db | head; tail -n 1 $out/bench_prof.json | cut -c1-600
