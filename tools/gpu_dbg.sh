#!/bin/bash
# usage (from the repository root): tools/gpu_dbg.sh "<flag list>" "<columns>"
# runs the diagnostic build (python -m btrblocks_amd.build --debug-steps), never the product library
export BTR_LIB_PATH=$PWD/btrblocks_amd/libbtrblocks_amd_dbg.so
mkdir -p gpurun_out
for f in $1; do
BTR_DEBUG_FLAGS=$f timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline --no-verify --no-plan-leg --no-copy-ceiling --columns $2 > gpurun_out/dbg_$f.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/dbg_$f.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("flags=$f", {k:v["kernel_ms"] for k,v in d["roofline"]["per_column"].items()})
    for x in open("gpurun_out/dbg_$f.log"):
        if x.startswith("[btr debug]"): print("  ", x.strip())
else: print("flags=$f failed", open("gpurun_out/dbg_$f.log").read()[-300:])
PY
done
