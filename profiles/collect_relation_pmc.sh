#!/bin/bash
# usage (on the GPU box, from the repository root): profiles/collect_relation_pmc.sh <tag>
# HBM traffic of every column launch of the mixed relation: two counter passes (FETCH_SIZE, WRITE_SIZE; each in
# a run of its own) over `bench.py --relation --serial` (one launch per column, in column order).
# profiles/summarize_relation_pmc.py <tag> adds the result to profiles/pmc_traffic.json ("relation").
set -o pipefail
tag=$1
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_rel_$c -o ${tag} -- python3 bench.py --relation --serial --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-verify --no-copy-ceiling > $out/${tag}_rel_$c.log 2>&1 || { tail -5 $out/${tag}_rel_$c.log; exit 1; }
done
find $out/${tag}_rel_FETCH_SIZE $out/${tag}_rel_WRITE_SIZE -name "*counter_collection.csv"
