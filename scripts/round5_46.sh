#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r5an; mkdir -p $out
python3 bench.py --dtype fp32 --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo rc $?
grep "bench.py" $out/bench.err | tail -4
python3 - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print("fp32", d["ms_per_step"], d["step_ms"]["median"]); x=d["ddp_1rank"]; print({k:x[k] for k in x if k.startswith(("ms_","plain","overhead"))}); print(json.dumps(x.get("fresh_process")))
PY
