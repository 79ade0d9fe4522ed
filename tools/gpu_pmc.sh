#!/bin/bash
# usage (GPU box, repository root): tools/gpu_pmc.sh <tag> <columns> <counter> [<counter> ...]
# One counter pass (rocprofv3 --pmc) over bench columns; prints the mean per launch of every btr_ kernel.
# Library under test: BTR_LIB_PATH (default: the product library).
set -o pipefail
tag=$1; cols=$2; shift 2
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/${tag}_pmc -o ${tag} -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-verify --no-plan-leg --no-copy-ceiling --columns $cols > $out/${tag}_pmc.log 2>&1 || { tail -5 $out/${tag}_pmc.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/${tag}_pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if not k.startswith("btr_"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print("$tag", k)
    for c, v in sorted(d.items()): print(f"   {c:40s} {v / cnt[(k, c)]:16.0f}")
PY
