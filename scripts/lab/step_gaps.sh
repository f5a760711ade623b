#!/bin/bash
# Where is the GPU idle inside a steady-state training step?  Kernel trace of a few steps; lists the idle gaps between
# consecutive kernels (all streams merged) by size, with the kernels on either side, and the idle time by position in the step.
# Usage: bash scripts/lab/step_gaps.sh [bf16|fp32] [steps]
export TMPDIR=/tmp; DT=${1:-bf16}; N=${2:-8}; out=gpurun_out/r2gaps_$DT; mkdir -p $out
rocprofv3 --output-format csv --kernel-trace -d $out/prof -o st -- python3 scripts/lab/step_few.py $DT $N > $out/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
opt = [i for i, n in enumerate(names) if "FusedAdam" in n or ("multi_tensor_apply" in n and "adam" in n.lower())]
# step boundaries: first adam kernel of each step (adam kernels come in bursts)
bursts = [opt[0]] + [b for a, b in zip(opt, opt[1:]) if b - a > 50]
print("steps seen", len(bursts))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
lo, hi = bursts[-4], bursts[-1]          # three full steps: from the optimiser of step k to the optimiser of step k+3
ss = rows[lo:hi]
t0 = int(ss[0]["Start_Timestamp"]); t1 = int(ss[-1]["Start_Timestamp"])
# merge intervals (two streams overlap)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), i) for i, r in enumerate(ss))
gaps, cur_end, cur_i, busy = [], iv[0][1], iv[0][2], 0
start = iv[0][0]
for s, e, i in iv[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_i, i, cur_end - t0))
        busy += cur_end - start; start = s
    if e > cur_end:
        cur_end, cur_i = e, i
busy += cur_end - start
n_steps = 3
print(f"3 steps: wall {(t1-t0)/1e6/n_steps:.2f} ms/step, busy {busy/1e6/n_steps:.2f} ms/step, idle {sum(g[0] for g in gaps)/1e6/n_steps:.2f} ms/step in {len(gaps)/n_steps:.0f} gaps/step")
for thr in (5e3, 2e4, 1e5):
    sel = [g for g in gaps if g[0] >= thr]
    print(f"  gaps >= {thr/1e3:.0f} us: {len(sel)/n_steps:.0f} per step, {sum(g[0] for g in sel)/1e6/n_steps:.2f} ms/step")
print("--- largest gaps (us, position in the 3-step window in ms, before -> after)")
for g, a, b, pos in sorted(gaps, reverse=True)[:40]:
    print(f"{g/1e3:8.1f} us @ {pos/1e6:7.2f} ms  {short(ss[a]['Kernel_Name'])}  ->  {short(ss[b]['Kernel_Name'])}")
# idle by 1-ms bins of the first step in the window
step_len = (t1 - t0) / n_steps
bins = [0.0] * (int(step_len / 1e6) + 2)
for g, a, b, pos in gaps:
    if pos < step_len:
        bins[int(pos / 1e6)] += g / 1e3
print("--- idle us per 1-ms bin of one step:", [int(v) for v in bins])
PY
find $out/prof -type f -size +2M -delete
