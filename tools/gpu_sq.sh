#!/bin/bash
# usage (GPU box, repository root): tools/gpu_sq.sh <tag> <columns>
# One SQ counter pass over the bench columns: where the waves' cycles go (parked, issue-stalled, active).
set -o pipefail
tag=$1; cols=$2
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $out/${tag}_sq -o ${tag} -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-verify --no-plan-leg --no-copy-ceiling --columns $cols > $out/${tag}_sq.log 2>&1 || { tail -5 $out/${tag}_sq.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/${tag}_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if not k.startswith("btr_"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    n = cnt[(k, "SQ_WAVE_CYCLES")]
    print(k, "launches", n)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {v / n:16.0f}  {v / d['SQ_WAVE_CYCLES']:.3f} of wave cycles")
PY
