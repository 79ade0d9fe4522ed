#!/bin/bash
# usage (on the GPU box, from the repository root): profiles/collect.sh <tag>
# 1. the default bench line; 2. rocprofv3 kernel trace + stats of the same command;
# 3. two counter passes (FETCH_SIZE, WRITE_SIZE; each in a run of its own) over the bench columns,
#    the PFOR column and a pure fill (onevalue) and a pure copy (uncompressed) whose byte counts are known.
# Summaries go to gpurun_out/<tag>_*; profiles/summarize.py turns them into the tracked files.
set -o pipefail
tag=$1
out=gpurun_out
mkdir -p $out
timeout -k 10 600 python3 -u bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -5 $out/${tag}_bench.err; exit 1; }
tail -c 600 $out/${tag}_bench.json
# the same columns beside each other (btr_gpu_decode_columns): a line of its own, its launches overlap
timeout -k 10 300 python3 -u bench.py --concurrent --no-cpu-baseline --no-pipeline > $out/${tag}_concurrent_bench.json 2> $out/${tag}_concurrent.err || { tail -5 $out/${tag}_concurrent.err; exit 1; }
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -o ${tag} -- python3 bench.py --no-cpu-baseline --no-pipeline --no-plan-leg --no-copy-ceiling > $out/${tag}_prof.log 2>&1 || { tail -5 $out/${tag}_prof.log; exit 1; }
cols=rle,dict,for,onevalue,uncompressed,pseudodecimal,pfor
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -o ${tag} -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --no-verify --no-plan-leg --no-copy-ceiling --columns $cols > $out/${tag}_pmc_fetch.log 2>&1 || { tail -5 $out/${tag}_pmc_fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -o ${tag} -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --no-verify --no-plan-leg --no-copy-ceiling --columns $cols > $out/${tag}_pmc_write.log 2>&1 || { tail -5 $out/${tag}_pmc_write.log; exit 1; }
find $out/${tag}_prof $out/${tag}_pmc_fetch $out/${tag}_pmc_write -name "*.csv" | head -20
