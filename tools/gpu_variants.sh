#!/bin/bash
# usage (from the repository root): tools/gpu_variants.sh "<variant names>" "<bench columns>"
# Times experiment builds (python -m btrblocks_amd.build --variant NAME -D...) against the product library.
mkdir -p gpurun_out
for v in product $1; do
  if [ $v = product ]; then unset BTR_LIB_PATH; else export BTR_LIB_PATH=$PWD/btrblocks_amd/libbtrblocks_amd_$v.so; fi
  timeout -k 10 200 python -u bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-pipeline --no-verify --no-plan-leg --no-copy-ceiling --columns $2 > gpurun_out/var_$v.log 2>&1 || { tail -5 gpurun_out/var_$v.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/var_$v.log") if x.startswith("{")]
d=json.loads(l[-1]); print("$v", {k:(v["kernel_ms"], v["kernel_ms_min"], v["frac"]) for k,v in d["roofline"]["per_column"].items()})
PY
done
