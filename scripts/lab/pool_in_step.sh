#!/bin/bash
# Why is the pooling forward slower inside the training step?  Kernel trace of a few steps, single- and dual-stream:
# duration of every pool launch, what ran concurrently with it, what ran right before it.
export TMPDIR=/tmp; out=gpurun_out/r2q; mkdir -p $out
for D in 1 0; do
  OMNIHD_DUAL_STREAM=$D rocprofv3 --output-format csv --kernel-trace -d $out/t$D -o kt -- python3 scripts/lab/step_few.py bf16 8 > $out/t$D.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$out/t$D/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
pools = [e for e in ev if "k_pool_fwd_lean2" in e[2]][-8:]
bwd = [e for e in ev if "k_pool_bwd_patch" in e[2]][-8:]
print("dual_stream=$D  pool fwd durations (us):", [round((e[1]-e[0])/1e3, 1) for e in pools], " bwd:", [round((e[1]-e[0])/1e3, 1) for e in bwd])
for s, e, n, q in pools[-3:]:
    conc = [(x[2][:50], round((min(e, x[1]) - max(s, x[0]))/1e3, 1), x[3]) for x in ev if x[0] < e and x[1] > s and x[2] != n]
    prev = [x for x in ev if x[1] <= s][-2:]
    print("   stream", q, "concurrent:", conc[:6], "| before:", [(p[2][:40], round((s - p[1])/1e3, 1)) for p in prev])
PY
  find $out/t$D -type f -size +2M -delete
done
