#!/usr/bin/env python3
"""Device-resident BtrBlocks decompression benchmark.

    python bench.py --gpus N --steps K --warmup W

N = 1 (default): four synthetic columns of `--rows` values each (default 268 435 456 = 4096 blocks
of 65 536 rows): int32 RLE(BP,BP), DICT(BP), FOR(BP) (1 GiB raw each) and a decimal-like double
column forcing PSEUDODECIMAL(BP,BP,UNCOMPRESSED) (2 GiB raw). One step = one decode pass over all
columns with the compressed chunks already resident in HBM.

N > 1: ONE mixed relation of 8 int32 + 4 double columns x `--rows` rows (16 GiB raw; nested
cascades of depth 3) whose blocks are sharded over the ranks as contiguous block ranges balanced
by decoded bytes (btrblocks_amd.shard.shard_ranges); every rank generates, compresses, uploads and
decodes only its own range, through ONE decode plan over the chunks of all its columns; no
data-path collective. The relation does not depend on N (it is
generated in segments of 512 blocks with fixed seeds), so scaling is strong and `value` is the
whole relation's decoded bytes over the slowest rank's time.
Launch: `python bench.py --gpus N` spawns the N ranks itself (one process per GPU) when it was
not started by torch.distributed.run; under torch.distributed.run it reads RANK / LOCAL_RANK /
WORLD_SIZE from the environment.

Prints ONE JSON line: metric/value (GB/s of uncompressed output, whole job), `roofline`
(algorithmic bytes = compressed bytes the decoder needs + decoded bytes written, per launch,
over the HIP-event duration of the decode kernel) and `cpu_baseline` (the CPU oracle's restated
decompression-speed harness on a bounded sample of rank 0's chunks, on rank 0's host cores).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
BLOCK_ROWS = 65536
SEGMENT_BLOCKS = 512   # the sharded relation is generated in segments of this many blocks

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
    "pfor_few": ("int_outliers_few", "pfor", 10),
    "pseudo_dictexp": ("double_decimal", "pseudodecimal(bp,dict(bp),uncompressed)", 4),
    "pseudo_clean": ("double_decimal_clean", "pseudodecimal(bp,bp,uncompressed)", 4),
    "pseudo_mixed": ("double_decimal_mixed", "pseudodecimal(bp,bp,uncompressed)", 4),
    "pseudo_rle": ("double_decimal_runs", "pseudodecimal(rle(dict(bp),bp),rle(bp,bp),uncompressed)", 11),
    "pseudo_auto": ("double_decimal", "pseudodecimal", 4),  # children picked by the compressor: (BP, DICT(BP), raw patches)
    "dbl_dict": ("double_dict", "dict(bp)", 12),
    "dbl_rle": ("double_runs", "rle(uncompressed,bp)", 13),
    # engine ceilings: a pure fill (ONE_VALUE) and a pure copy (UNCOMPRESSED)
    "onevalue": ("int_constant", "onevalue", 5),
    "uncompressed": ("int_for", "uncompressed", 6),
}
SINGLE_GPU_COLUMNS = "rle,dict,for,pseudodecimal"
RELATION_COLUMNS = "rle,dict,for,rle_dict,for_rle,freq_dict,pfor,onevalue,pseudodecimal,pseudo_rle,dbl_dict,dbl_rle"


def column_elem_bytes(name):
    return 8 if COLUMNS[name][0].startswith("double") else 4


def rank_block_range(names, rows, rank, world):
    """Block range of this rank: contiguous, balanced by the decoded bytes of all columns."""
    from btrblocks_amd import shard
    blocks = (rows + BLOCK_ROWS - 1) // BLOCK_ROWS
    block_rows = np.full(blocks, BLOCK_ROWS, dtype=np.int64)
    block_rows[-1] = rows - (blocks - 1) * BLOCK_ROWS
    per_row = sum(column_elem_bytes(n) for n in names)
    return shard.shard_ranges(block_rows * per_row, world)[rank]


def generate_range(name, rows, b0, b1, segmented):
    """Rows of blocks [b0, b1) of column `name`. Segmented: the column is the concatenation of
    segments of SEGMENT_BLOCKS blocks, segment s generated with seed + 1000 * s, so any rank can
    produce its range without the rest of the column."""
    from btrblocks_amd import synth
    gen, _, seed = COLUMNS[name]
    fn = getattr(synth, gen)
    r0, r1 = b0 * BLOCK_ROWS, min(rows, b1 * BLOCK_ROWS)
    if not segmented:
        return fn(rows, seed=seed)[r0:r1]
    seg_rows = SEGMENT_BLOCKS * BLOCK_ROWS
    parts = []
    for s in range(r0 // seg_rows, (r1 + seg_rows - 1) // seg_rows):
        s0 = s * seg_rows
        n = min(seg_rows, rows - s0)
        v = fn(n, seed=seed + 1000 * s)
        parts.append(v[max(r0, s0) - s0:min(r1, s0 + n) - s0])
    return np.ascontiguousarray(np.concatenate(parts)) if len(parts) != 1 else np.ascontiguousarray(parts[0])


def build_columns(names, rows, b0, b1, segmented, threads, valid_fraction=None):
    import btrblocks_amd as B
    from btrblocks_amd import synth
    cols = []
    for name in names:
        _, spec, seed = COLUMNS[name]
        t0 = time.time()
        values = generate_range(name, rows, b0, b1, segmented)
        t1 = time.time()
        nullmap = None if valid_fraction is None else synth.random_nullmap(values.size, valid_fraction, seed=seed + 77 + b0)
        image = B.compress_column(values, nullmap, spec, threads=threads)
        t2 = time.time()
        cols.append({"name": name, "spec": spec, "image": image, "raw_bytes": values.nbytes, "rows": values.size,
                     "gen_s": t1 - t0, "compress_s": t2 - t1})
        del values
    return cols


def cpu_baseline(cols, sample_chunks, budget_s=12.0):
    """Restated decompression-speed (oracle) over the first `sample_chunks` chunks of each column:
    all host threads (the reference's TBB pool) and one thread, in GB/s (1e9) and in the
    reference's own unit (decompressed MiB per second, decompression-speed.cpp:205-238)."""
    import btrblocks_amd as B
    from oracle import pyoracle
    total_s, total_bytes, threads = 0.0, 0, 0
    one_s, one_bytes = 0.0, 0
    per = {}
    for c in cols:
        chunks = B.part_chunks(c["image"])[:sample_chunks]
        sec, nbytes, threads = pyoracle.decompression_speed(c["image"], chunks, threads=0, reps=1)
        reps = int(max(1, min(40, (0.7 * budget_s / len(cols)) / max(sec, 1e-6))))
        sec, nbytes, threads = pyoracle.decompression_speed(c["image"], chunks, threads=0, reps=reps)
        per[c["name"]] = round(nbytes / sec / 1e9, 3)
        total_s += sec
        total_bytes += nbytes
        few = chunks[:max(1, sample_chunks // 8)]
        sec1, nbytes1, _ = pyoracle.decompression_speed(c["image"], few, threads=1, reps=1)
        reps1 = int(max(1, min(10, (0.3 * budget_s / len(cols)) / max(sec1, 1e-6))))
        sec1, nbytes1, _ = pyoracle.decompression_speed(c["image"], few, threads=1, reps=reps1)
        one_s += sec1
        one_bytes += nbytes1
    return {
        "value": round(total_bytes / total_s / 1e9, 3),
        "unit": "GB/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {sample_chunks} chunks (65536 rows each) of every column, CPU oracle restating "
                  f"decompression-speed (one task per chunk, {threads} std::threads, warm-up + timed passes); "
                  f"one-thread figure on the first {max(1, sample_chunks // 8)} chunks",
        "MiB_per_s": round(total_bytes / total_s / 2**20, 1),
        "one_thread": {"value": round(one_bytes / one_s / 1e9, 3), "unit": "GB/s", "cores": 1,
                       "MiB_per_s": round(one_bytes / one_s / 2**20, 1)},
        "per_column": per,
    }


def pipeline_rate(ctx, cols, passes=2, slots=3):
    """Host-to-host rate of the same columns through the part pipeline (btr_gpu_pipeline): the
    compressed columns start as 16 MiB column parts in pinned host memory (a warm page cache), the
    decoded columns end in pinned host memory. Every pass walks the headers and compiles the decode
    programs again, copies the parts in, decodes and copies the columns out, overlapped over parts."""
    import btrblocks_amd as B
    from btrblocks_amd import api, files
    part_views, out_need, total_in = [], 0, 0
    layouts = []
    for c in cols:
        image = c["image"]
        groups, cur, cur_bytes = [], [], 0
        for off, n in B.part_chunks(image):
            if cur and cur_bytes + n > files.PART_BYTES:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append((off, n))
            cur_bytes += n
        if cur:
            groups.append(cur)
        col_parts = []
        for g in groups:
            at = 4 + 4 * len(g)
            places = []
            for off, n in g:
                at = (at + 15) // 16 * 16
                places.append(at)
                at += n
            col_parts.append((g, places, at))
            total_in += (at + 255) // 256 * 256
        layouts.append(col_parts)
    pinned_in = api.PinnedBuffer(total_in + 256)
    at = 0
    jobs = []  # (part view, decoded bytes)
    for c, col_parts in zip(cols, layouts):
        image = c["image"]
        elem = 8 if image[B.part_chunks(image)[0][0] + 2] == 1 else 4
        col_out = 0
        for g, places, nbytes in col_parts:
            view = pinned_in.array[at:at + nbytes]
            view[:4 + 4 * len(g)] = np.array([len(g)] + places, dtype=np.uint32).view(np.uint8)
            decoded = 0
            for (off, n), place in zip(g, places):
                view[place:place + n] = image[off:off + n]
                tuples = int(image[off + 8:off + 12].view(np.uint32)[0])
                decoded = (decoded + 15) // 16 * 16 + tuples * elem
            jobs.append((view, decoded, col_out))
            col_out = (col_out + decoded + 15) // 16 * 16
            at += (nbytes + 255) // 256 * 256
        out_need = max(out_need, col_out)
    pinned_out = api.PinnedBuffer(out_need + 256)
    pipe = api.PartPipeline(ctx, slots)
    total_out = sum(d for _, d, _ in jobs)

    def one_pass():
        for view, decoded, col_out in jobs:
            pipe.submit(view, host_dst=pinned_out.ptr + col_out, dst_bytes=decoded)
        pipe.drain()

    one_pass()  # grows the pipeline's buffers
    before = pipe.stats
    c0, w0 = before.compile_seconds, before.wait_seconds
    t0 = time.perf_counter()
    for _ in range(passes):
        one_pass()
    dt = (time.perf_counter() - t0) / passes
    after = pipe.stats
    res = {
        "what": "pinned host column parts -> plan compile -> H2D -> decode -> D2H -> pinned host columns, overlapped over parts",
        "out_GBps": round(total_out / dt / 1e9, 2),
        "in_GBps": round(sum(v.size for v, _, _ in jobs) / dt / 1e9, 2),
        "ms_per_pass": round(dt * 1e3, 2),
        "parts": len(jobs),
        "slots": slots,
        "compile_s_per_pass": round((after.compile_seconds - c0) / passes, 4),
        "slot_wait_s_per_pass": round((after.wait_seconds - w0) / passes, 4),
    }
    pipe.close()
    pinned_in.close()
    pinned_out.close()
    return res


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=268435456,
                    help="rows per column (N = 1) / rows of the sharded relation (N > 1)")
    ap.add_argument("--columns", default=None,
                    help=f"default: {SINGLE_GPU_COLUMNS} at N = 1, the mixed relation ({RELATION_COLUMNS}) at N > 1")
    ap.add_argument("--relation", action="store_true", help="N = 1: decode the mixed relation of the N > 1 runs")
    ap.add_argument("--cpu-sample-chunks", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the host-to-host leg through the part pipeline")
    ap.add_argument("--nulls", type=float, default=None, metavar="P",
                    help="give every column a random nullmap with a fraction P of valid rows and decode the byte-per-row maps too")
    ap.add_argument("--concurrent", action="store_true",
                    help="decode the columns of a step beside each other (btr_gpu_decode_columns); default at N > 1, "
                         "where a rank's launches are too small to fill its GPU one at a time")
    ap.add_argument("--beside-leg", action="store_true",
                    help="N = 1: after the timed region, time the same step with the columns beside each other as well "
                         "(columns_beside_each_other; its launches overlap, so leave it out of profiled runs)")
    ap.add_argument("--no-copy-ceiling", action="store_true", help="skip the device-to-device copy rate in the roofline object")
    ap.add_argument("--no-plan-leg", action="store_true",
                    help="N = 1: skip the secondary leg that decodes the step through one plan over all columns")
    ap.add_argument("--serial", action="store_true", help="N > 1: one column launch after the other, as at N = 1")
    ap.add_argument("--per-column-plans", action="store_true",
                    help="mixed relation: one plan per column, decoded beside each other, instead of ONE plan over the "
                         "chunks of all columns of the rank's block range (launches = kernel kinds, longest first)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of a sample of blocks after the timed region")
    ap.add_argument("--verify-stride", type=int, default=257, help="check every N-th chunk (and the last one)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--init-dist", action="store_true",
                    help="N = 1 rehearsal: initialise a one-rank process group (and use its barrier) as a rank of an N > 1 run does")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: this process never touches the GPU,
    starts one child per rank, forwards rank 0's JSON line and fails if any rank fails."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def verify_sample(reader, stride):
    """Every stride-th chunk and the last one against the CPU oracle, bit for bit."""
    from oracle import pyoracle
    n = len(reader.chunks)
    checked = 0
    for i in sorted(set(list(range(0, n, stride)) + [n - 1])):
        o, ln = reader.chunks[i]
        want = pyoracle.decode_chunk(reader.image[o:o + ln])
        lo = int(reader.out_offsets[i])
        got = reader.d_out[lo:lo + want.nbytes].cpu().numpy()
        if not np.array_equal(got, want.view(np.uint8)):
            raise AssertionError(f"chunk {i} differs from the oracle")
        checked += 1
    return checked


def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.one_device:
        local_rank = 0
    if world > 1 or args.init_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # rehearsal: a one-rank group, so that the process holds RCCL's streams like a rank of an N > 1 run
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")  # where the scalar reductions live

    import btrblocks_amd as B
    from btrblocks_amd import api

    relation = world > 1 or args.relation
    names = [n.strip() for n in (args.columns or (RELATION_COLUMNS if relation else SINGLE_GPU_COLUMNS)).split(",") if n.strip()]
    b0, b1 = rank_block_range(names, args.rows, rank, world)
    host_threads = max(1, (os.cpu_count() or 8) // max(1, world))
    host_threads = min(host_threads, int(os.environ.get("OMP_NUM_THREADS", host_threads)))
    cols = build_columns(names, args.rows, b0, b1, relation, host_threads, args.nulls)
    want_nm = args.nulls is not None

    ctx = B.GpuContext(local_rank)
    readers = []
    t_open = 0.0
    for c in cols:
        t0 = time.time()
        r = B.BtrReader(c["image"], ctx=ctx)  # envelopes in one call + upload of the image
        t_open += time.time() - t0
        readers.append(r)
    torch.cuda.synchronize()
    t0 = time.time()
    plans = [r.plan(None, want_nm) for r in readers]
    t_plan = time.time() - t0
    stats = [p.stats for p in plans]
    out_bytes = sum(s.decoded_bytes for s in stats)
    alg_bytes = sum(s.decoded_bytes + s.needed_bytes for s in stats)
    stream = api.current_stream_ptr()

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or args.init_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # The mixed relation is handed over whole: one plan over the chunks of ALL columns of this rank's
    # block range, as decompression-speed.cpp:45-58 gives every column x part x chunk to one scheduler.
    # Its launches are kernel kinds (longest first), not columns.
    rel_plan = None
    if relation and not args.serial and not args.concurrent and not args.per_column_plans and len(readers) > 1:
        t0 = time.time()
        descs = [r.chunk_desc(i, want_nm) for r in readers for i in range(len(r.chunks))]
        rel_plan = api.DecodePlan(ctx, descs, keepalive=tuple(readers))
        t_rel_plan = time.time() - t0
        rel_stats = rel_plan.stats
    concurrent = rel_plan is not None or ((args.concurrent or world > 1) and not args.serial and len(plans) > 1)

    def step(events=None):
        if rel_plan is not None:
            rel_plan.decode(stream)
            return
        if concurrent:  # the columns beside each other; per-launch events would time overlapped launches
            api.decode_columns(ctx, plans, stream)
            return
        for i, p in enumerate(plans):
            if events is not None:
                events[i][0].record()
            p.decode(stream)
            if events is not None:
                events[i][1].record()

    for _ in range(args.warmup):
        step()
    barrier()
    for p in plans + ([rel_plan] if rel_plan is not None else []):  # malformed-stream flags of the warm-up passes
        p.check(stream)

    evs = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plans]
           for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    for p in plans + ([rel_plan] if rel_plan is not None else []):
        p.check(stream)
    verified = 0
    if not args.no_verify:
        verified = sum(verify_sample(r, args.verify_stride) for r in readers)

    my_ms = elapsed / args.steps * 1e3
    if concurrent:
        # per-launch durations of the same plans, one launch after the other, outside the timed region
        barrier()
        for k in range(args.steps):
            for i, p in enumerate(plans):
                evs[k][i][0].record()
                p.decode(stream)
                evs[k][i][1].record()
        barrier()
    beside_ms = None
    if args.beside_leg and not concurrent and world == 1 and len(plans) > 1:
        # secondary figure: the same step with the columns beside each other
        for _ in range(2):
            api.decode_columns(ctx, plans, stream)
        barrier()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            api.decode_columns(ctx, plans, stream)
        barrier()
        beside_ms = (time.perf_counter() - t2) / args.steps * 1e3
        for p in plans:
            p.check(stream)
    one_plan_ms = None
    if world == 1 and rel_plan is None and not concurrent and len(readers) > 1 and not args.no_plan_leg:
        # secondary figure: the same step through ONE plan over the chunks of all columns (launches per kernel,
        # beside each other); its launches overlap, so it is not the timed region of this line
        descs = [r.chunk_desc(i, want_nm) for r in readers for i in range(len(r.chunks))]
        one_plan = api.DecodePlan(ctx, descs, keepalive=tuple(readers))
        for _ in range(2):
            one_plan.decode(stream)
        barrier()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            one_plan.decode(stream)
        barrier()
        one_plan_ms = (time.perf_counter() - t2) / args.steps * 1e3
        one_plan.check(stream)
        one_plan_launches = int(one_plan.stats.launches)
        one_plan.close()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(out_bytes), float(verified)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_out, verified_all = float(tot[0].item()), int(tot[1].item())
        per_rank = [torch.zeros(3, dtype=torch.float64, device=red_dev) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([my_ms, float(b0), float(b1)], dtype=torch.float64, device=red_dev))
        per_rank = [{"rank": r, "ms_per_step": round(float(v[0]), 4), "blocks": [int(v[1]), int(v[2])]}
                    for r, v in enumerate(per_rank)]
    else:
        total_out, verified_all = float(out_bytes), verified
        per_rank = [{"rank": 0, "ms_per_step": round(my_ms, 4), "blocks": [b0, b1]}]

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
            "alg_bytes_per_value": round(ab / max(1, s.tuples), 3),
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

    copy_ceiling = None
    if rank == 0 and not args.no_copy_ceiling:
        # what the box's own device-to-device copy reaches (read + written bytes): context for the fraction of
        # the 8 TB/s spec peak that `roofline.frac` is quoted against
        try:
            a = torch.empty(1 << 28, dtype=torch.int32, device=dev)
            b = torch.empty_like(a)
            for _ in range(3):
                b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            copy_ceiling = round(10 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del a, b
        except Exception:
            copy_ceiling = None
    if rank == 0:
        # HBM bytes per launch of the dominant kernel: not collected in this run (counters need rocprofv3
        # passes of their own, profiles/collect.sh) but read from the tracked summary of those passes. The
        # passes ran the N = 1 default columns; for any other launch size the kernel's measured ratio of
        # traffic to algorithmic bytes is applied to this launch's algorithmic bytes, and the source says so.
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                summary = json.load(open(tpath))
                kern = per_column[cols[dom]["name"]]["kernels"][0]
                tag = summary.get("tag", "untagged")
                detail = summary.get("detail", {}).get(kern)
                rel_cols = summary.get("relation", {}).get("columns", {})
                if relation and cols[dom]["name"] in rel_cols:  # the counter passes of the mixed relation, per column
                    detail = dict(rel_cols[cols[dom]["name"]], column=cols[dom]["name"])
                    tag = summary["relation"].get("tag", tag)
                    if detail.get("algorithmic_bytes") == dom_alg:
                        traffic = detail["read_bytes"] + detail["write_bytes"]
                        traffic_source = (f"profiles/pmc_traffic.json ({tag}, relation): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                          "passes of this column's launch")
                        detail = None
                if traffic is not None:
                    pass
                elif detail is not None and detail.get("algorithmic_bytes") == dom_alg:
                    traffic = summary["bytes_per_launch"][kern]
                    traffic_source = f"profiles/pmc_traffic.json ({tag}): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload"
                elif detail is not None:
                    traffic = int(round(detail["traffic_over_algorithmic"] * dom_alg))
                    traffic_source = (f"profiles/pmc_traffic.json ({tag}): {kern} moved {detail['traffic_over_algorithmic']} x its "
                                      f"algorithmic bytes in the counter passes of column '{detail['column']}' at 4096 blocks per column; "
                                      "that ratio times this launch's algorithmic bytes")
            except Exception:
                traffic, traffic_source = None, None
        dtypes = sorted({"f64" if column_elem_bytes(n) == 8 else "int32" for n in names})
        line = {
            "metric": "device-resident decompression GB/s (uncompressed output)",
            "value": round(total_out * args.steps / elapsed / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "+".join(dtypes),
            "data": "synthetic",
            "config": {
                "workload": (f"mixed relation of {len(names)} columns x {args.rows} rows "
                             f"({','.join(c['spec'] for c in cols)}), blocks sharded over {world} GPU(s) as contiguous ranges"
                             if relation else
                             f"{len(names)} synthetic columns x {args.rows} rows ({','.join(c['spec'] for c in cols)})")
                            + ", 65536-row blocks, compressed chunks resident in HBM, one workgroup per block",
                "rows": args.rows,
                "blocks": int((args.rows + BLOCK_ROWS - 1) // BLOCK_ROWS),
                "total_output_bytes": int(total_out),
                "rank0_output_bytes": int(out_bytes),
                "sharding": "independent block ranges per GPU, no collective",
                "launches": ("ONE plan over the chunks of all columns of the rank's block range (launches = kernel kinds, longest "
                             "first, beside each other); roofline.per_column timed one column launch after the other outside "
                             "the timed region" if rel_plan is not None else
                             "columns beside each other (btr_gpu_decode_columns); roofline.per_column timed one launch after "
                             "the other outside the timed region" if concurrent else "one column launch after the other"),
                "verified_chunks": verified_all,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "peak_source": "MI355X HBM3E specification (8 TB/s); frac = achieved / peak",
                "device_copy_GBps": copy_ceiling,  # torch device-to-device copy of 1 GiB on this GPU, read + written bytes
                "kernel": ",".join(per_column[cols[dom]["name"]]["kernels"]),
                "column": cols[dom]["name"],
                "alg_bytes_per_launch": int(dom_alg),
                "avg_launch_ms": round(float(mean_ms[dom]), 4),
                "all_launches": {"achieved": round(achieved_all, 1), "frac": round(achieved_all / HBM_PEAK_GBS, 4),
                                 "launches_per_step": int(sum(s.launches for s in stats)),
                                 "alg_bytes_per_step": int(alg_bytes), "kernel_ms_per_step": round(float(mean_ms.sum()), 4)},
                "per_column": per_column,
            },
            "per_rank": per_rank,
            "host": {
                "open_s": round(t_open, 3),          # part inspection + H2D upload of the compressed images
                "plan_s": round(t_plan, 3),          # btr_gpu_plan_create_part, all columns
                "plan_compile_s": round(sum(s.compile_seconds for s in stats), 4),
                "plan_upload_s": round(sum(s.upload_seconds for s in stats), 4),
                "chunks": int(sum(s.chunks for s in stats)),
                "compress_s": round(sum(c["compress_s"] for c in cols), 2),
                "generate_s": round(sum(c["gen_s"] for c in cols), 2),
                "threads": host_threads,
            },
        }
        if one_plan_ms is not None:
            line["one_plan_over_all_columns"] = {"ms_per_step": round(one_plan_ms, 4), "GBps": round(total_out / one_plan_ms / 1e6, 2),
                                                 "launches_per_step": one_plan_launches,
                                                 "what": "the same step through one btr_gpu_plan over the chunks of all columns "
                                                         "(launches per kernel, beside each other), timed after the timed region; "
                                                         "not the value of this line"}
        if beside_ms is not None:
            line["columns_beside_each_other"] = {"ms_per_step": round(beside_ms, 4),
                                                 "GBps": round(total_out / beside_ms / 1e6, 2),
                                                 "what": "the same step through btr_gpu_decode_columns; not the value of this line"}
        if world == 1 and not args.no_pipeline:
            line["host"]["pipeline"] = pipeline_rate(ctx, cols)
        if world > 1:
            # the same relation on ONE GPU, from the tracked profile run of `bench.py --relation` (the N = 1 default of
            # this script is another workload, configs 2 + 3): the figure a strong-scaling efficiency is taken against
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_config4_bench.json")), reverse=True):
                try:
                    one = json.load(open(path))
                    if one.get("n_gpus") == 1 and one["config"]["rows"] == args.rows and \
                            one["config"]["total_output_bytes"] == int(total_out):
                        line["same_relation_on_one_gpu"] = {"value": one["value"], "unit": "GB/s", "ms_per_step": one["ms_per_step"],
                                                            "source": "profiles/" + os.path.basename(path)}
                        break
                except Exception:
                    continue
        if rel_plan is not None:
            line["relation_plan"] = {"chunks": int(rel_stats.chunks), "launches_per_step": int(rel_stats.launches),
                                     "kernels": kernel_names(rel_stats.kernel_mask), "plan_s": round(t_rel_plan, 3),
                                     "compile_s": round(rel_stats.compile_seconds, 4)}
        if not args.no_cpu_baseline:  # rank 0's host cores, on rank 0's own chunks
            line["cpu_baseline"] = cpu_baseline(cols, min(args.cpu_sample_chunks, max(8, 3072 // len(cols))))
        print(json.dumps(line), flush=True)
    if world > 1 or args.init_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()
