"""Multi-rank path on CPU (gloo, world_size 2): block ranges shard without overlap or gaps, each
rank decodes only its shard (oracle stands in for the GPU here) and the size/offset exchange
reassembles the column."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, rows, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import btrblocks_amd as B
    from btrblocks_amd import shard, synth
    from oracle import pyoracle
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    values = synth.int_rle(rows, seed=5)                 # every rank knows the column (host side)
    image = B.compress_column(values, None, "rle(bp,bp)", threads=1)
    chunks = B.part_chunks(image)
    sizes = [B.chunk_info(image[o:o + n]).decoded_bytes for o, n in chunks]
    first, end = shard.shard_ranges(sizes, world)[rank]
    mine = [pyoracle.decode_chunk(image[o:o + n]) for o, n in chunks[first:end]]
    part = np.concatenate(mine) if mine else np.zeros(0, np.int32)
    (row_off, byte_off), (tot_rows, tot_bytes) = shard.exchange_offsets(part.size, part.nbytes)
    ok = np.array_equal(part, values[row_off:row_off + part.size]) and tot_rows == rows and byte_off == 4 * row_off
    q.put((rank, first, end, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_and_reassemble():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rows = 5 * 65536 + 1234
    procs = [ctx.Process(target=_worker, args=(r, 2, port, rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, _, _, ok in results)
    assert results[0][1] == 0 and results[0][2] == results[1][1] and results[1][2] == 6


def test_shard_ranges_balance():
    from btrblocks_amd import shard
    sizes = [262144] * 4095 + [1000]
    for world in (1, 2, 4, 8):
        r = shard.shard_ranges(sizes, world)
        assert r[0][0] == 0 and r[-1][1] == len(sizes)
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        per = [sum(sizes[a:b]) for a, b in r]
        assert max(per) - min(per) <= 2 * 262144


def test_bench_relation_does_not_depend_on_the_rank_count(monkeypatch):
    """bench.py shards ONE relation: the rows a rank generates for its block range are the rows
    any other split of the same relation holds at those positions."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "SEGMENT_BLOCKS", 3)
    rows = 10 * 65536 + 777
    names = ["rle", "pseudodecimal", "dict"]
    whole = {n: bench.generate_range(n, rows, 0, 11, True) for n in names}
    assert all(v.size == rows for v in whole.values())
    for world in (2, 3, 4, 8):
        ranges = [bench.rank_block_range(names, rows, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == 11
        assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        for n in names:
            got = np.concatenate([bench.generate_range(n, rows, b0, b1, True) for b0, b1 in ranges])
            assert np.array_equal(got.view(np.uint8), whole[n].view(np.uint8)), (n, world)


def test_bench_parent_spawns_one_process_per_rank(monkeypatch, capsys):
    """`python bench.py --gpus N` outside torch.distributed.run: the parent starts N children with
    RANK / LOCAL_RANK / WORLD_SIZE set, forwards rank 0's line and fails when a rank fails."""
    sys.path.insert(0, ROOT)
    import subprocess
    import bench
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None):
            self.rank = int(env["RANK"])
            started.append((self.rank, env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"], stdout))
            self.returncode = 3 if (self.rank == 2 and fail) else 0

        def communicate(self):
            return (b'{"n_gpus": 4}\n', None)

        def wait(self):
            return self.returncode

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    args = bench.parse_args()
    fail = False
    assert bench.spawn_ranks(args) == 0
    assert [s[0] for s in started] == [0, 1, 2, 3] and all(s[2] == "4" and s[3] == "127.0.0.1" for s in started)
    assert started[0][4] == subprocess.PIPE and started[1][4] == subprocess.DEVNULL
    assert '"n_gpus": 4' in capsys.readouterr().out
    fail = True
    assert bench.spawn_ranks(args) == 1
