#!/usr/bin/env python3
"""usage: tools/trace_timeline.py <kernel_trace.csv> <launches per step> [first step] [steps]
Prints start / end of every btr_ kernel of a step, relative to the step's first start (rocprofv3 --kernel-trace)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
first = int(sys.argv[3]) if len(sys.argv) > 3 else 5
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:34], r["Queue_Id"], r["Stream_Id"],
             r["Grid_Size_X"], r["Workgroup_Size_X"]) for r in rows if r["Kernel_Name"].startswith("btr_"))
for s in range(first, first + steps):
    grp = ks[s * n:(s + 1) * n]
    t0 = min(g[0] for g in grp)
    print("step", s, "span %.1f us" % ((max(g[1] for g in grp) - t0) / 1e3))
    for g in grp:
        print("   %-34s q=%s st=%s wgs=%5d start %7.1f end %7.1f dur %7.1f" % (
            g[2], g[3], g[4], int(g[5]) // int(g[6]), (g[0] - t0) / 1e3, (g[1] - t0) / 1e3, (g[1] - g[0]) / 1e3))
