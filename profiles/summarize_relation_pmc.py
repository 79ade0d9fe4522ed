#!/usr/bin/env python3
"""Adds the per-column HBM traffic of the mixed relation (profiles/collect_relation_pmc.sh <tag>) to
profiles/pmc_traffic.json under "relation": the launches of `bench.py --relation --serial` come in
column order, so launch k of the run belongs to column k mod (number of columns)."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")


def launches(counter):
    f = glob.glob(os.path.join(out, f"{tag}_rel_{counter}", "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("btr_")]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"], float(r["Counter_Value"])) for r in rows]


line = json.loads([l for l in open(os.path.join(out, f"{tag}_rel_FETCH_SIZE.log")) if l.startswith("{")][-1])
cols = list(line["roofline"]["per_column"].items())
fetch, write = launches("FETCH_SIZE"), launches("WRITE_SIZE")
assert len(fetch) == len(write) and len(fetch) % len(cols) == 0, (len(fetch), len(write), len(cols))
rel = {}
for k, (name, c) in enumerate(cols):
    f = [v for (kn, v) in fetch[k::len(cols)]]
    w = [v for (kn, v) in write[k::len(cols)]]
    kernel = fetch[k][0]
    assert kernel in c["kernels"], (name, kernel, c["kernels"])
    rd, wr = 2 * 1024 * sum(f) / len(f), 1024 * sum(w) / len(w)
    alg = c["needed_bytes"] + c["decoded_bytes"]
    rel[name] = {"kernel": kernel, "read_bytes": round(rd), "write_bytes": round(wr), "launches": len(f),
                 "algorithmic_bytes": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 4),
                 "write_over_decoded": round(wr / c["decoded_bytes"], 4)}
path = os.path.join(root, "profiles", "pmc_traffic.json")
summary = json.load(open(path))
summary["relation"] = {"tag": tag, "rows": line["config"]["rows"], "columns": rel}
json.dump(summary, open(path, "w"), indent=1)
print(json.dumps(summary["relation"], indent=1))
