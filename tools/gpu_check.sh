#!/bin/bash
# usage (from the repository root): tools/gpu_check.sh <tag> [bench args]   -- parity tests, then bench (plain + root-skipped)
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/parity_$tag.log 2>&1
rc=$?; tail -3 gpurun_out/parity_$tag.log
[ $rc -ne 0 ] && exit $rc
for f in 0 2; do
BTR_DEBUG_FLAGS=$f timeout -k 10 300 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --columns rle,dict,for,pseudodecimal "$@" > gpurun_out/bench_${tag}_$f.log 2>&1 || { tail -5 gpurun_out/bench_${tag}_$f.log; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/bench_${tag}_$f.log") if x.startswith("{")]
d=json.loads(l[-1]); print("flags=$f", {k:(v["kernel_ms"], v["frac"]) for k,v in d["roofline"]["per_column"].items()})
PY
done
