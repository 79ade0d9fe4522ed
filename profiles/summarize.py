#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of profiles/collect.sh <tag> (under gpurun_out/) into the tracked
summaries: <tag>_kernel_stats.csv, <tag>_bench_line.json and pmc_traffic.json.

HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB-sized units of 1024 bytes):
FETCH_SIZE counts 128-byte read requests as 64 bytes on gfx950 for wide streaming reads, so it is
doubled; WRITE_SIZE is taken as reported. The pure fill (ONE_VALUE) and pure copy (UNCOMPRESSED)
columns, whose byte counts are known, are kept in the summary as a check of both conventions.
"""
import csv, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")


def per_kernel(path, counter):
    """mean counter value per launch and kernel, plus the launch sequence (for the calibration columns)"""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("btr_")]
    by = defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return by, rows


fetch, frows = per_kernel(os.path.join(out, f"{tag}_pmc_fetch", f"{tag}_counter_collection.csv"), "FETCH_SIZE")
write, wrows = per_kernel(os.path.join(out, f"{tag}_pmc_write", f"{tag}_counter_collection.csv"), "WRITE_SIZE")

bench = json.loads(open(os.path.join(out, f"{tag}_bench.json")).read().strip().splitlines()[-1])
cols = dict(bench["roofline"]["per_column"])
try:  # the counter runs decode more columns than the default line (PFOR, fill, copy): take their byte counts too
    pmc_line = [l for l in open(os.path.join(out, f"{tag}_pmc_fetch.log")) if l.startswith("{")][-1]
    for name, c in json.loads(pmc_line)["roofline"]["per_column"].items():
        cols.setdefault(name, c)
except (OSError, IndexError, KeyError, ValueError):
    pass
summary = {"tag": tag, "unit": "bytes", "convention": "2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 per launch", "bytes_per_launch": {},
           "detail": {}}
# The generic integer kernel decodes three columns in the counter runs (rle, onevalue, uncompressed), launched
# in that order each step: split its launches by position.
order = {"btr_decode_int_kernel": ["rle", "onevalue", "uncompressed"]}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    names = order.get(k, [None])
    for j, nm in enumerate(names):
        fj, wj = f[j::len(names)], w[j::len(names)]
        if not fj or not wj:
            continue
        rd = 2 * 1024 * sum(fj) / len(fj)
        wr = 1024 * sum(wj) / len(wj)
        key = k if nm in (None, "rle") else f"{k}[{nm}]"
        summary["detail"][key] = {"read_bytes": round(rd), "write_bytes": round(wr), "launches": len(fj)}
        if nm in (None, "rle"):
            summary["bytes_per_launch"][k] = round(rd + wr)
for name, c in cols.items():
    for k in c["kernels"]:
        key = k if f"{k}[{name}]" not in summary["detail"] else f"{k}[{name}]"
        if key in summary["detail"] and "column" not in summary["detail"][key]:
            det = summary["detail"][key]
            det["algorithmic_bytes"] = c["needed_bytes"] + c["decoded_bytes"]
            det["column"] = name
            det["traffic_over_algorithmic"] = round((det["read_bytes"] + det["write_bytes"]) / det["algorithmic_bytes"], 4)
            det["write_over_decoded"] = round(det["write_bytes"] / c["decoded_bytes"], 4)
json.dump(summary, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(out, f"{tag}_prof", f"{tag}_kernel_stats.csv"), os.path.join(prof, f"{tag}_kernel_stats_rocprofv3.csv"))
json.dump(bench, open(os.path.join(prof, f"{tag}_bench_line.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
