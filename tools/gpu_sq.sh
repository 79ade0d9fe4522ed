#!/bin/bash
# usage (from the repository root): tools/gpu_sq.sh <columns>  -- SQ instruction counters of the decode kernels of one bench run
export TMPDIR=/tmp
out=gpurun_out
rm -rf $out/sq
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/sq -o sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --columns $1 > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen=set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if not k.startswith("btr_"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key=(k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[k]+=1
for k, d in acc.items():
    print(k, "dispatches", cnt[k], {c: round(v / cnt[k] / 1e6, 2) for c, v in d.items()})
PY
