"""GPU: a workgroup that is handed a launch entry or a decode program other than the one the host
compiled must end with BTR_ERR_CORRUPT (status bit ST_PROGRAM = 0x80) in bounded time instead of
interpreting it: every loop bound, table offset and barrier count of the kernels comes out of the
program. The reference's counterpart of a failed decode is an exception
(btrblocks/scheme/integer/PBP.cpp:44-47). The plan's device tables are damaged through the
btr_gpu_plan_debug_access test hook; after each case the table is restored and the same plan
decodes bit-exactly again."""
import ctypes as C
import struct
import time

import numpy as np
import pytest
import torch

import btrblocks_amd as B
from btrblocks_amd import _native as N
from btrblocks_amd import api, synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu

ST_PROGRAM = 0x80
ITEM_BYTES, ENTRY_BYTES = 64, 16          # sizeof(WorkItem), sizeof(LaunchEntry)
ITEM_PROG_OFF, ITEM_PROG_BYTES = 32, 36   # offsets inside a WorkItem


def _access(plan, region, offset, data=None, n=0):
    lib = plan.lib
    if data is None:
        buf = (C.c_uint8 * n)()
        N.check(lib.btr_gpu_plan_debug_access(plan.handle, region, offset, buf, n, 0))
        return bytes(buf)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    N.check(lib.btr_gpu_plan_debug_access(plan.handle, region, offset, buf, len(data), 1))


class _Plan:
    """Four small chunks of three kernel kinds (the generic integer kernel, the Pseudodecimal stream
    kernel and the PFOR kernel on side streams) in one plan."""

    def __init__(self, ctx):
        specs = [(synth.int_rle(1000), "rle(bp,bp)"), (synth.double_decimal(257), "pseudodecimal(bp,bp,uncompressed)"),
                 (np.arange(3000, dtype=np.int32) // 2, "rle(bp,bp)"), (synth.int_outliers(513), "pfor"),
                 (synth.int_for(65536), "for(bp)"), (synth.int_dict(65536), "dict(bp)")]
        self.chunks = [B.compress_chunk(v, None, s) for v, s in specs]
        self.want = [pyoracle.decode_chunk(c) for c in self.chunks]
        dev = torch.device("cuda", 0)
        self.srcs, self.outs, descs = [], [], []
        for c, w in zip(self.chunks, self.want):
            t = torch.zeros(c.size + 16, dtype=torch.uint8, device=dev)
            t[:c.size] = torch.from_numpy(c)
            o = torch.zeros(w.nbytes + 64, dtype=torch.uint8, device=dev)
            d = N.ChunkDesc()
            d.host_src, d.dev_src, d.src_bytes = c.ctypes.data, t.data_ptr(), c.size
            d.dev_dst, d.dst_bytes = o.data_ptr(), w.nbytes
            self.srcs.append(t), self.outs.append(o), descs.append(d)
        self.plan = api.DecodePlan(ctx, descs)
        self.stream = api.current_stream_ptr()

    def decode_status(self):
        for o in self.outs:
            o.zero_()
        t0 = time.time()
        self.plan.decode(self.stream)
        st = C.c_uint32()
        rc = self.plan.lib.btr_gpu_plan_check(self.plan.ctx.handle, self.plan.handle, self.stream, C.byref(st), None)
        assert time.time() - t0 < 20, "a damaged plan must end in bounded time"
        return rc, st.value

    def assert_clean(self):
        rc, st = self.decode_status()
        assert rc == 0 and st == 0
        for o, w in zip(self.outs, self.want):
            assert np.array_equal(o[:w.nbytes].cpu().numpy(), w.view(np.uint8))


def _entries(p):
    raw = _access(p.plan, 1, 0, n=ENTRY_BYTES * len(p.chunks))
    return [list(struct.unpack_from("<IIII", raw, ENTRY_BYTES * k)) for k in range(len(p.chunks))]


def test_damaged_launch_entries_and_programs_are_reported(gpu_ctx):
    p = _Plan(gpu_ctx)
    lib = p.plan.lib
    p.assert_clean()
    entries = _entries(p)
    for k, (item, off, size, check) in enumerate(entries):
        assert check == lib.btr_debug_launch_entry_check(item, off, size)
        original = struct.pack("<IIII", item, off, size, check)
        # 1. a program place the check word does not cover (a stale or mis-strided launch order)
        _access(p.plan, 1, ENTRY_BYTES * k, struct.pack("<IIII", item, off + 16, size, check))
        rc, st = p.decode_status()
        assert rc == 2 and st & ST_PROGRAM, f"entry {k}: wrong prog_off went unnoticed"
        _access(p.plan, 1, ENTRY_BYTES * k, original)
        p.assert_clean()
        # 2. an item index far outside the plan
        _access(p.plan, 1, ENTRY_BYTES * k, struct.pack("<IIII", 0x7FFFFFF0, off, size, check))
        rc, st = p.decode_status()
        assert rc == 2 and st & ST_PROGRAM
        _access(p.plan, 1, ENTRY_BYTES * k, original)
        # 3. a self-consistent entry that names another place than the work item does
        other = entries[(k + 1) % len(entries)]
        _access(p.plan, 1, ENTRY_BYTES * k,
                struct.pack("<IIII", item, other[1], other[2], lib.btr_debug_launch_entry_check(item, other[1], other[2])))
        rc, st = p.decode_status()
        assert rc == 2 and st & ST_PROGRAM
        _access(p.plan, 1, ENTRY_BYTES * k, original)
        # 4. a truncated program: entry and work item agree on a length the program header does not have
        short = size - 16
        _access(p.plan, 1, ENTRY_BYTES * k, struct.pack("<IIII", item, off, short, lib.btr_debug_launch_entry_check(item, off, short)))
        _access(p.plan, 0, ITEM_BYTES * item + ITEM_PROG_BYTES, struct.pack("<I", short))
        rc, st = p.decode_status()
        assert rc == 2 and st & ST_PROGRAM
        _access(p.plan, 1, ENTRY_BYTES * k, original)
        _access(p.plan, 0, ITEM_BYTES * item + ITEM_PROG_BYTES, struct.pack("<I", size))
        # 5. header fields of the program itself: step / pipe / sequence counts, table base, magic
        header = _access(p.plan, 2, off, n=32)
        for word, value in ((0, 0x40000000), (0, 0), (1, 0xFFFF), (2, 1 << 20), (3, 77), (4, 12), (5, 8), (6, 0)):
            bad = bytearray(header)
            struct.pack_into("<I", bad, 4 * word, value)
            _access(p.plan, 2, off, bytes(bad))
            rc, st = p.decode_status()
            assert rc == 2 and st & ST_PROGRAM, f"entry {k}: header word {word} = {value:#x} went unnoticed"
        _access(p.plan, 2, off, header)
        p.assert_clean()
    p.plan.close()


def test_multi_kind_plans_decoded_back_to_back_leave_the_device_idle(gpu_ctx):
    """A plan of several kernel kinds decoded many times (its launches spread over the context's
    streams), then closed; then plans decoded back to back on one stream, one of them with a small
    side launch, and a device-wide synchronisation. With side streams of raised priority (hardware
    queues beyond the process's usual four) that last synchronisation never returned."""
    p = _Plan(gpu_ctx)
    for _ in range(12):
        p.assert_clean()
    p.plan.close()
    torch.cuda.synchronize()
    specs = [("int_rle", "rle(bp,bp)"), ("int_dict", "dict(bp)"), ("int_for", "for(bp)"), ("double_decimal", "pseudodecimal"),
             ("int_outliers", "pfor"), ("int_rle_pool", "rle(dict(bp),bp)")]
    readers, columns = [], []
    for k, (gen, spec) in enumerate(specs):
        values = getattr(synth, gen)(3 * 65536 + 1000 * k)
        columns.append(values)
        readers.append(B.BtrReader(B.compress_column(values, None, spec), ctx=gpu_ctx))
    plans = [r.plan(None, False) for r in readers]
    assert any(pl.stats.launches > 1 for pl in plans)
    stream = api.current_stream_ptr()
    for _ in range(3):
        for pl in plans:
            pl.decode(stream)
    t0 = time.time()
    torch.cuda.synchronize()
    assert time.time() - t0 < 20
    for pl, r, values in zip(plans, readers, columns):
        assert pl.check(stream) == 0
        assert np.array_equal(r.column_host().view(np.uint8), values.view(np.uint8))
