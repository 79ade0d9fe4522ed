#!/bin/bash
# usage (on the GPU box, from the repository root): profiles/collect_relation.sh <tag>
# The 16 GiB mixed relation on one GPU (one relation-level plan; one column launch after the other; rocprofv3
# kernel stats of the latter), the eighth of it that one of eight ranks decodes (relation-level plan / one
# column launch after the other / per-column plans beside each other), and a two-rank rehearsal of the
# multi-GPU launch on this one device. Results go to gpurun_out/<tag>_*.
set -o pipefail
tag=$1
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python3 -u bench.py --relation --no-cpu-baseline --no-pipeline > $out/${tag}_config4_bench.json 2> $out/${tag}_config4.err || { tail -5 $out/${tag}_config4.err; exit 1; }
timeout -k 10 400 python3 -u bench.py --relation --serial --no-cpu-baseline --no-pipeline --no-verify > $out/${tag}_config4_serial.json 2>> $out/${tag}_config4.err || { tail -5 $out/${tag}_config4.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_config4_prof -o ${tag}_config4 -- python3 bench.py --relation --serial --steps 10 --no-cpu-baseline --no-pipeline --no-verify --no-copy-ceiling > $out/${tag}_config4_prof.log 2>&1 || { tail -5 $out/${tag}_config4_prof.log; exit 1; }
timeout -k 10 300 python3 -u bench.py --relation --rows 33554432 --steps 40 --no-pipeline > $out/${tag}_eighth_plan.json 2> $out/${tag}_eighth.err || { tail -5 $out/${tag}_eighth.err; exit 1; }
timeout -k 10 300 python3 -u bench.py --relation --rows 33554432 --steps 40 --serial --no-cpu-baseline --no-pipeline > $out/${tag}_eighth_serial.json 2>> $out/${tag}_eighth.err || { tail -5 $out/${tag}_eighth.err; exit 1; }
timeout -k 10 300 python3 -u bench.py --relation --rows 33554432 --steps 40 --concurrent --no-cpu-baseline --no-pipeline > $out/${tag}_eighth_concurrent.json 2>> $out/${tag}_eighth.err || { tail -5 $out/${tag}_eighth.err; exit 1; }
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --rows 67108864 --one-device --backend gloo > $out/${tag}_gpus2_rehearsal.json 2> $out/${tag}_gpus2.err || { tail -5 $out/${tag}_gpus2.err; exit 1; }
for f in config4_bench config4_serial eighth_plan eighth_serial eighth_concurrent gpus2_rehearsal; do tail -c 300 $out/${tag}_$f.json | head -c 10 > /dev/null; python3 - <<PY
import json
l=[x for x in open("$out/${tag}_$f.json") if x.startswith("{")]
d=json.loads(l[-1]); print("$f", d["value"], d["ms_per_step"], d["n_gpus"])
PY
done
