#!/bin/bash
# usage (from the repository root): tools/gpu_variants_rel.sh "<variant names>" <rows> [bench flags]
# The mixed relation (bench.py --relation) with experiment builds against the product library.
mkdir -p gpurun_out
variants=$1; rows=$2; shift 2 || true
for v in product $variants; do
  if [ $v = product ]; then unset BTR_LIB_PATH; else export BTR_LIB_PATH=$PWD/btrblocks_amd/libbtrblocks_amd_$v.so; fi
  timeout -k 10 300 python -u bench.py --relation --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --no-pipeline --no-verify "$@" > gpurun_out/rel_$v.log 2>&1 || { tail -5 gpurun_out/rel_$v.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/rel_$v.log") if x.startswith("{")]
d=json.loads(l[-1]); print("$v", d["value"], d["ms_per_step"], {k:v["kernel_ms"] for k,v in d["roofline"]["per_column"].items()})
PY
done
