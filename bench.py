#!/usr/bin/env python3
"""Device-resident BtrBlocks decompression benchmark.

    python bench.py --gpus N --steps K --warmup W

Workload (N=1): three synthetic int32 columns of `--rows` values each (default 268 435 456 = 1 GiB
raw, 4096 blocks of 65 536 rows), one forcing RLE(BP,BP), one DICT(BP), one FOR(BP). One step =
one decode pass over all three columns with the compressed chunks already resident in HBM.
With N>1 every rank decodes its own three columns (different seeds) on its own GPU: block ranges
shard with no data-path collective, so scaling is weak and `value` is the sum over ranks.

Prints ONE JSON line: metric/value (GB/s of uncompressed output, whole job), `roofline`
(algorithmic bytes = compressed bytes the decoder needs + decoded bytes written, per launch,
over the HIP-event duration of the decode kernel) and `cpu_baseline` (the CPU oracle's restated
decompression-speed harness on a bounded sample of the same chunks, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

COLUMNS = {
    # name: (generator, scheme spec, column seed)
    "rle": ("int_rle", "rle(bp,bp)", 1),
    "dict": ("int_dict", "dict(bp)", 2),
    "for": ("int_for", "for(bp)", 3),
    "pseudodecimal": ("double_decimal", "pseudodecimal(bp,bp,uncompressed)", 4),
    # nested cascades of the mixed-relation configuration (depth 3, trees the reference compressor emits)
    "rle_dict": ("int_rle_pool", "rle(dict(bp),bp)", 7),
    "for_rle": ("int_rle", "for(rle(bp,bp))", 8),
    "freq_dict": ("int_frequency_pool", "frequency(dict(bp))", 9),
    "pfor": ("int_outliers", "pfor", 10),
    "pseudo_rle": ("double_decimal_runs", "pseudodecimal(rle(dict(bp),bp),rle(bp,bp),uncompressed)", 11),
    "pseudo_auto": ("double_decimal", "pseudodecimal", 4),  # children picked by the compressor: (BP, DICT(BP), raw patches)
    "dbl_dict": ("double_dict", "dict(bp)", 12),
    "dbl_rle": ("double_runs", "rle(uncompressed,bp)", 13),
    # engine ceilings: a pure fill (ONE_VALUE) and a pure copy (UNCOMPRESSED)
    "onevalue": ("int_constant", "onevalue", 5),
    "uncompressed": ("int_for", "uncompressed", 6),
}


def build_columns(names, rows, rank, threads, valid_fraction=None):
    import btrblocks_amd as B
    from btrblocks_amd import synth
    cols = []
    for name in names:
        gen, spec, seed = COLUMNS[name]
        t0 = time.time()
        values = getattr(synth, gen)(rows, seed=seed + 1000 * rank)
        t1 = time.time()
        nullmap = None if valid_fraction is None else synth.random_nullmap(rows, valid_fraction, seed=seed + 77)
        image = B.compress_column(values, nullmap, spec, threads=threads)
        t2 = time.time()
        cols.append({"name": name, "spec": spec, "image": image, "raw_bytes": values.nbytes,
                     "gen_s": t1 - t0, "compress_s": t2 - t1})
        del values
    return cols


def cpu_baseline(cols, sample_chunks, budget_s=12.0):
    """Restated decompression-speed (oracle) over the first `sample_chunks` chunks of each column."""
    import btrblocks_amd as B
    from oracle import pyoracle
    total_s, total_bytes, threads = 0.0, 0, 0
    per = {}
    for c in cols:
        chunks = B.part_chunks(c["image"])[:sample_chunks]
        sec, nbytes, threads = pyoracle.decompression_speed(c["image"], chunks, threads=0, reps=1)
        reps = int(max(1, min(40, (budget_s / len(cols)) / max(sec, 1e-6))))
        sec, nbytes, threads = pyoracle.decompression_speed(c["image"], chunks, threads=0, reps=reps)
        per[c["name"]] = round(nbytes / sec / 1e9, 3)
        total_s += sec
        total_bytes += nbytes
    return {
        "value": round(total_bytes / total_s / 1e9, 3),
        "unit": "GB/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {sample_chunks} chunks (65536 rows each) of every column, CPU oracle restating "
                  f"decompression-speed (one task per chunk, {threads} std::threads, warm-up + timed passes)",
        "per_column": per,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=268435456, help="rows per column")
    ap.add_argument("--columns", default="rle,dict,for")
    ap.add_argument("--cpu-sample-chunks", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nulls", type=float, default=None, metavar="P",
                    help="give every column a random nullmap with a fraction P of valid rows and decode the byte-per-row maps too")
    ap.add_argument("--verify", action="store_true", help="compare a sample of chunks with the oracle")
    ap.add_argument("--verify-stride", type=int, default=257, help="with --verify: check every N-th chunk")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.one_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")  # where the two scalar reductions live

    import btrblocks_amd as B
    from btrblocks_amd import api

    names = [n.strip() for n in args.columns.split(",") if n.strip()]
    host_threads = max(1, (os.cpu_count() or 8) // max(1, world))
    host_threads = min(host_threads, int(os.environ.get("OMP_NUM_THREADS", host_threads)) if world == 1 else host_threads)
    cols = build_columns(names, args.rows, rank, host_threads, args.nulls)
    want_nm = args.nulls is not None

    ctx = B.GpuContext(local_rank)
    readers = []
    t_plan = 0.0
    for c in cols:
        t0 = time.time()
        r = B.BtrReader(c["image"], ctx=ctx)
        r.plan(None, want_nm)
        t_plan += time.time() - t0
        readers.append(r)
    plans = [r.plan(None, want_nm) for r in readers]
    stats = [p.stats for p in plans]
    out_bytes = sum(s.decoded_bytes for s in stats)
    alg_bytes = sum(s.decoded_bytes + s.needed_bytes for s in stats)
    stream = api.current_stream_ptr()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(events=None):
        for i, p in enumerate(plans):
            if events is not None:
                events[i][0].record()
            p.decode(stream)
            if events is not None:
                events[i][1].record()

    for _ in range(args.warmup):
        step()
    barrier()
    for p in plans:  # malformed-stream flags of the warm-up passes
        p.check(stream)
    if args.verify:
        from oracle import pyoracle
        for r in readers:
            host = r.column_host()
            pos = 0
            for i, (o, n) in enumerate(r.chunks):
                rows = r.infos[i].tuple_count
                if i % args.verify_stride == 0:
                    want = pyoracle.decode_chunk(r.image[o:o + n])
                    assert np.array_equal(host[pos:pos + rows].view(np.uint8), want.view(np.uint8)), f"chunk {i} differs"
                pos += rows

    evs = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plans]
           for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    for p in plans:
        p.check(stream)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(out_bytes)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_out = float(tot.item())
    else:
        total_out = float(out_bytes)

    # per-launch kernel durations from HIP events on the launch stream
    kern_ms = np.array([[a.elapsed_time(b) for (a, b) in evs[k]] for k in range(args.steps)])  # steps x columns
    mean_ms = kern_ms.mean(axis=0)
    per_column = {}
    for i, c in enumerate(cols):
        s = stats[i]
        ab = s.decoded_bytes + s.needed_bytes
        per_column[c["name"]] = {
            "scheme": c["spec"],
            "kernel_ms": round(float(mean_ms[i]), 4),
            "kernel_ms_min": round(float(kern_ms[:, i].min()), 4),
            "out_GBps": round(s.decoded_bytes / mean_ms[i] / 1e6, 1),
            "alg_GBps": round(ab / mean_ms[i] / 1e6, 1),
            "frac": round(ab / mean_ms[i] / 1e6 / HBM_PEAK_GBS, 4),
            "needed_bytes": int(s.needed_bytes),
            "stored_bytes": int(s.stored_bytes),
            "decoded_bytes": int(s.decoded_bytes),
            "alg_bytes_per_value": round(ab / s.tuples, 3),
            "lds_bytes": int(s.max_lds_bytes),
        }
    # kernel behind each plan (one launch per plan in this workload)
    from btrblocks_amd import _native
    lib = _native.load()

    def kernel_names(mask):
        return [lib.btr_kernel_name(k).decode() for k in range(32) if mask >> k & 1]

    for i, c in enumerate(cols):
        per_column[c["name"]]["kernels"] = kernel_names(stats[i].kernel_mask)
    # the dominant kernel: the launch that takes the largest share of a step
    dom = int(np.argmax(mean_ms))
    dom_alg = stats[dom].decoded_bytes + stats[dom].needed_bytes
    achieved = dom_alg / (mean_ms[dom] / 1e3) / 1e9          # GB/s of the dominant kernel's launches
    achieved_all = alg_bytes / (mean_ms.sum() / 1e3) / 1e9   # GB/s over all launches of a step

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("bytes_per_launch", {}).get(
                    per_column[cols[dom]["name"]]["kernels"][0])
            except Exception:
                traffic = None
        line = {
            "metric": "device-resident decompression GB/s (uncompressed output)",
            "value": round(total_out * args.steps / elapsed / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32" if all(c["raw_bytes"] == 4 * args.rows for c in cols) else "int32+f64",
            "data": "synthetic",
            "config": {
                "workload": f"{len(names)} synthetic columns x {args.rows} rows ({','.join(c['spec'] for c in cols)}), "
                            f"65536-row blocks, compressed chunks resident in HBM, one workgroup per block",
                "rows_per_column": args.rows,
                "blocks_per_column": int(stats[0].chunks),
                "per_gpu_output_bytes": int(out_bytes),
                "sharding": "independent block ranges per GPU, no collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel": ",".join(per_column[cols[dom]["name"]]["kernels"]),
                "column": cols[dom]["name"],
                "alg_bytes_per_launch": int(dom_alg),
                "avg_launch_ms": round(float(mean_ms[dom]), 4),
                "all_launches": {"achieved": round(achieved_all, 1), "frac": round(achieved_all / HBM_PEAK_GBS, 4),
                                 "launches_per_step": int(sum(s.launches for s in stats)),
                                 "alg_bytes_per_step": int(alg_bytes), "kernel_ms_per_step": round(float(mean_ms.sum()), 4)},
                "per_column": per_column,
            },
            "host": {
                "plan_build_s": round(t_plan, 3),
                "compress_s": round(sum(c["compress_s"] for c in cols), 2),
                "generate_s": round(sum(c["gen_s"] for c in cols), 2),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cols, args.cpu_sample_chunks)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
