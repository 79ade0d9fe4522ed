#!/usr/bin/env python3
"""Host-to-host rate of the bench workload: compressed column images start in pinned host memory,
decoded columns end in pinned host memory. Per pass: H2D copy of every image, decode, D2H copy of
every decoded column, on one stream (serial) and with the three stages of different columns
overlapped on three streams. Prints one JSON line (GB/s of decoded output)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import btrblocks_amd as B
from btrblocks_amd import api

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 268435456
names = ["rle", "dict", "for"]
cols = bench.build_columns(names, rows, 0, 16)
ctx = B.GpuContext(0)
readers = [B.BtrReader(c["image"], ctx=ctx) for c in cols]
plans = [r.plan(None, False) for r in readers]
h_in = [torch.from_numpy(r.image).pin_memory() for r in readers]
h_out = [torch.empty(r.d_out.numel(), dtype=torch.uint8).pin_memory() for r in readers]
out_bytes = sum(p.stats.decoded_bytes for p in plans)
in_bytes = sum(t.numel() for t in h_in)


def serial():
    s = api.current_stream_ptr()
    for r, p, hi, ho in zip(readers, plans, h_in, h_out):
        r.d_image.copy_(hi, non_blocking=True)
        p.decode(s)
        ho.copy_(r.d_out, non_blocking=True)
    torch.cuda.synchronize()


streams = [torch.cuda.Stream() for _ in readers]


def overlapped():
    for r, p, hi, ho, st in zip(readers, plans, h_in, h_out, streams):
        with torch.cuda.stream(st):
            r.d_image.copy_(hi, non_blocking=True)
            p.decode(st.cuda_stream)
            ho.copy_(r.d_out, non_blocking=True)
    torch.cuda.synchronize()


res = {}
for name, fn in (("serial", serial), ("overlapped", overlapped)):
    fn()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    dt = (time.perf_counter() - t0) / 5
    res[name] = {"ms": round(dt * 1e3, 2), "out_GBps": round(out_bytes / dt / 1e9, 1)}
for p in plans:
    p.check(api.current_stream_ptr())
print(json.dumps({"rows_per_column": rows, "compressed_bytes": in_bytes, "decoded_bytes": out_bytes, **res}))
