#!/bin/bash
# usage (from the repository root): tools/gpu_dbg.sh "<flag list>" "<columns>"
for f in $1; do
BTR_DEBUG_FLAGS=$f timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --columns $2 > gpurun_out/dbg_$f.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/dbg_$f.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("flags=$f", {k:v["kernel_ms"] for k,v in d["roofline"]["per_column"].items()})
else: print("flags=$f failed", open("gpurun_out/dbg_$f.log").read()[-300:])
PY
done
