#!/bin/bash
# kernel-trace profile of the fusion training step; prints the top kernels per step
OUT=${1:-gpurun_out/prof_step}; export TMPDIR=/tmp; mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o step -- python3 scripts/try_fusion.py r1 bf16 > $OUT/run.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
steps=13.0
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step ~ %.1f (13 steps incl. first)"%(tot/1e6/steps))
for r in rows[:40]:
    print("%6.2f%% calls/step %7.1f avg_us %9.1f ms/step %6.2f  %s" % (float(r["Percentage"]), int(r["Calls"])/steps, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6/steps, r["Name"][:100]))
PY
find $OUT -type f -size +3M -delete
