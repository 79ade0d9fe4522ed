#!/bin/bash
# usage (from the repository root): tools/gpu_cols.sh <tag> "<pytest -k expression>" "<bench columns>"  -- parity subset, then bench
tag=$1
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py tests/test_gpu_reader.py -m gpu -x -q --timeout 200 --timeout-method thread -k "$2" > gpurun_out/${tag}_parity.log 2>&1
rc=$?; tail -2 gpurun_out/${tag}_parity.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert|Mismatch" gpurun_out/${tag}_parity.log | head -20; exit $rc; }
timeout -k 10 300 python -u bench.py --columns $3 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/${tag}_bench.json") if l.startswith("{")][-1])
print({k:(v["kernel_ms"],v["kernel_ms_min"],v["frac"],v["lds_bytes"]) for k,v in d["roofline"]["per_column"].items()})
PY
