#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5am; mkdir -p $out
python3 bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo rc $?
python3 - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("fp32", d["ms_per_step"], d["step_ms"]["median"]); x=d["ddp_1rank"]; print({k:x[k] for k in x if k.startswith(("ms_","plain","overhead"))}); print(x["step_ms"])
PY
