#!/bin/bash
# usage (from the repository root): tools/gpu_alone.sh "<columns>"   -- every column in a bench run of its own
for c in $1; do
timeout -k 10 200 python -u bench.py --steps 20 --warmup 3 --no-cpu-baseline --columns $c > gpurun_out/alone_$c.log 2>&1 || { tail -3 gpurun_out/alone_$c.log; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/alone_$c.log") if x.startswith("{")]
d=json.loads(l[-1]); print({k:(v["kernel_ms"], v["kernel_ms_min"], v["frac"], v["kernels"]) for k,v in d["roofline"]["per_column"].items()})
PY
done
