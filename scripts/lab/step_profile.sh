#!/bin/bash
# Steady-state kernel statistics of the bf16 training step (MIOpen db seeded, immediate find lookups): launches per step
# and GPU time by kernel.  Usage: bash scripts/lab/step_profile.sh [bf16|fp32] [steps]
export TMPDIR=/tmp; DT=${1:-bf16}; N=${2:-10}; out=${STEP_PROFILE_OUT:-gpurun_out/r3v_$DT}; mkdir -p $out
rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof -o st -- python3 scripts/lab/step_few.py $DT $N > $out/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# steady-state region: the last N steps = launches after the (3+...) warm-up; split by the optimizer kernel count
names = [r["Kernel_Name"] for r in rows]
opt = [i for i, n in enumerate(names) if "multi_tensor_apply" in n and "adam" in n.lower()]
print("total launches", len(rows), "adam launches", len(opt))
# take the last $N steps: find boundaries by the last adam kernel of each step (fused adamw = a few launches per step)
per_step = max(1, len(opt) // ($N + int('${STEP_PROFILE_WARM:-3}')))
cut = opt[-per_step * $N - 1] + 1 if len(opt) > per_step * $N else 0
ss = rows[cut:]
t0, t1 = int(ss[0]["Start_Timestamp"]), int(ss[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ss)
print(f"steady region: {len(ss)} launches over $N steps = {len(ss)/$N:.0f} per step; wall {(t1-t0)/1e6/$N:.2f} ms/step; sum of kernel durations {busy/1e6/$N:.2f} ms/step")
ours = lambda n: "omnihd::" in n or "_ZN6omnihd" in n
t_ours = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ss if ours(r["Kernel_Name"]))
n_ours = sum(1 for r in ss if ours(r["Kernel_Name"]))
print(f"hand-written (libomnihd_hip.so) kernels: {t_ours/1e6/$N:.2f} ms/step in {n_ours/$N:.0f} launches/step; library / framework kernels (MIOpen, rocBLAS, ATen): "
      f"{(busy-t_ours)/1e6/$N:.2f} ms/step in {(len(ss)-n_ours)/$N:.0f} launches/step")
agg = collections.defaultdict(lambda: [0, 0])
for r in ss:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
    agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("--- by GPU time")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t/1e6/$N:7.3f} ms/step {c/$N:7.1f} calls/step  {k}")
import os
probe = os.environ.get("STEP_PROFILE_PROBE")           # print the neighbours of the first steady-state launch matching this
if probe:
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:150]
    hits = [i for i, r in enumerate(ss) if probe in r["Kernel_Name"]][:2]
    for i in hits:
        print("--- neighbours of", probe)
        for j in range(max(0, i - 5), min(len(ss), i + 4)):
            r = ss[j]
            print(("  >> " if j == i else "     ") + f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3:8.1f} us  grid {r.get('Grid_Size', '?')}  {short(r['Kernel_Name'])}")
pairs = os.environ.get("STEP_PROFILE_PAIRS")           # histogram of (previous kernel, next kernel) around every launch matching this
if pairs:
    short2 = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    hist = collections.defaultdict(lambda: [0, 0])
    for i, r in enumerate(ss):
        if pairs in r["Kernel_Name"] and 0 < i < len(ss) - 1:
            key = (short2(ss[i - 1]["Kernel_Name"]), r.get("Grid_Size", "?"), short2(ss[i + 1]["Kernel_Name"]))
            hist[key][0] += 1; hist[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("--- (previous, grid, next) around", pairs)
    for k, (c, t) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{c/$N:6.1f} x {t/1e3/max(c,1):7.1f} us  prev {k[0]} | grid {k[1]} | next {k[2]}")
if os.environ.get("STEP_PROFILE_GAPS"):                 # when is no kernel running at all?  (union of the kernel intervals over all streams)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), i) for i, r in enumerate(ss))
    busy_u, cur_s, cur_e, cur_i = 0, iv[0][0], iv[0][1], iv[0][2]
    gaps = []
    for s_, e_, i in iv[1:]:
        if s_ > cur_e:
            busy_u += cur_e - cur_s
            gaps.append((s_ - cur_e, cur_i, i))
            cur_s, cur_e, cur_i = s_, e_, i
        elif e_ > cur_e:
            cur_e, cur_i = e_, i
    busy_u += cur_e - cur_s
    short3 = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    print(f"--- GPU idle: some kernel running {busy_u/1e6/$N:.2f} ms/step, none {(t1-t0-busy_u)/1e6/$N:.2f} ms/step in {len(gaps)/$N:.0f} gaps/step")
    hist = collections.defaultdict(lambda: [0, 0])
    for g, a_, b_ in gaps:
        key = (short3(ss[a_]["Kernel_Name"]), short3(ss[b_]["Kernel_Name"]))
        hist[key][0] += 1; hist[key][1] += g
    for k, (c, t) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{t/1e6/$N:7.3f} ms/step {c/$N:6.1f} x {t/1e3/c:7.1f} us  after {k[0]} | before {k[1]}")
print("--- by launch count")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
    print(f"{c/$N:7.1f} calls/step {t/1e6/$N:7.3f} ms/step  {k}")
PY
find $out/prof -type f -size +2M -delete
