"""GPU: the allocation-free entry point (btr_gpu_decode_chunks_noalloc) against
btr_gpu_plan_create + btr_gpu_decode on the same chunks, bit for bit, on two calls in a row (the
second with fewer and smaller chunks: the static area keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

import btrblocks_amd as B
from btrblocks_amd import _native as N
from btrblocks_amd import api, synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _chunks(specs):
    return [B.compress_chunk(v, nm, spec) for v, nm, spec in specs]


def _decode_both(ctx, chunks, want_nullmap):
    import torch
    dev = torch.device("cuda", 0)
    infos = [B.chunk_info(c) for c in chunks]
    srcs = []
    for c in chunks:
        t = torch.zeros(c.size + 16, dtype=torch.uint8, device=dev)
        t[:c.size] = torch.from_numpy(c)
        srcs.append(t)

    def buffers():
        outs = [torch.full((max(16, int(i.decoded_bytes)) + 64,), 0xA5, dtype=torch.uint8, device=dev) for i in infos]
        nms = [torch.full((max(16, i.tuple_count) + 64,), 0xA5, dtype=torch.uint8, device=dev) for i in infos]
        descs = []
        for c, s, i, o, m in zip(chunks, srcs, infos, outs, nms):
            d = N.ChunkDesc()
            d.host_src, d.dev_src, d.src_bytes = c.ctypes.data, s.data_ptr(), c.size
            d.dev_dst, d.dst_bytes = o.data_ptr(), int(i.decoded_bytes)
            d.dev_nullmap_dst = m.data_ptr() if want_nullmap else None
            descs.append(d)
        return outs, nms, descs

    stream = api.current_stream_ptr()
    # the existing entry points
    outs_a, nms_a, descs_a = buffers()
    plan = api.DecodePlan(ctx, descs_a)
    plan.decode(stream)
    plan.check(stream)
    # the allocation-free one
    outs_b, nms_b, descs_b = buffers()
    arr = (N.ChunkDesc * len(descs_b))(*descs_b)
    st = C.c_uint32(1)
    N.check(ctx.lib.btr_gpu_decode_chunks_noalloc(ctx.handle, arr, len(descs_b), stream, C.byref(st)))
    assert st.value == 0
    torch.cuda.synchronize()
    for k, (a, b, ma, mb) in enumerate(zip(outs_a, outs_b, nms_a, nms_b)):
        assert torch.equal(a, b), f"chunk {k}: column differs from the planned decode"
        assert torch.equal(ma, mb), f"chunk {k}: nullmap differs from the planned decode"
        assert bool((b[int(infos[k].decoded_bytes):] == 0xA5).all())
    plan.close()
    got = [o[:int(i.decoded_bytes)].cpu().numpy() for o, i in zip(outs_b, infos)]
    # ... and against the CPU oracle, every row (null or not) and the byte-per-row nullmap
    for k, (c, g, i) in enumerate(zip(chunks, got, infos)):
        want, want_nm = pyoracle.decode_chunk(c, want_nullmap=True)
        assert np.array_equal(g, want.view(np.uint8)), f"chunk {k}: column differs from the oracle"
        if want_nullmap:
            assert np.array_equal(nms_b[k][:i.tuple_count].cpu().numpy(), want_nm), f"chunk {k}: nullmap differs from the oracle"
    return got


def test_noalloc_matches_planned_decode_twice(gpu_ctx):
    n = 65536
    big = [
        (synth.int_rle(n), None, "rle(bp,bp)"),
        (synth.int_dict(n), synth.random_nullmap(n, 0.8), "dict(bp)"),
        (synth.int_for(n), None, "for(bp)"),
        (synth.double_decimal(n), None, "pseudodecimal(bp,bp,uncompressed)"),
        (synth.int_outliers(n), None, "pfor"),
        (np.random.default_rng(15).integers(0, 2**31, n).astype(np.int32), None, "dict(bp)"),  # dictionary read from global memory
        (np.arange(n, dtype=np.int32) // 2, None, "rle(bp,bp)"),  # 32768 runs: tables spill to the scratch slice
        (synth.double_dict(n), synth.random_nullmap(n, 0.5), "dict(bp)"),
        (synth.double_decimal_runs(n), None, "pseudodecimal(rle(dict(bp),bp),rle(bp,bp),uncompressed)"),
        (synth.int_frequency(n), None, "frequency(bp)"),
    ] * 3
    first = _chunks(big)
    got = _decode_both(gpu_ctx, first, want_nullmap=True)
    for (v, nm, spec), g in zip(big, got):
        valid = np.ones(v.size, bool) if nm is None else nm.astype(bool)
        assert np.array_equal(g.view(v.dtype)[valid], v[valid]), spec
    # second call: fewer, smaller chunks
    small = [
        (synth.int_rle(1000), None, "rle(bp,bp)"),
        (synth.double_decimal(257), None, "pseudodecimal(bp,bp,uncompressed)"),
        (np.arange(3000, dtype=np.int32) // 2, None, "rle(bp,bp)"),
        (synth.int_outliers(513), None, "pfor"),
    ]
    got = _decode_both(gpu_ctx, _chunks(small), want_nullmap=False)
    for (v, nm, spec), g in zip(small, got):
        assert np.array_equal(g.view(v.dtype), v), spec


def test_noalloc_limits_and_corruption(gpu_ctx):
    import torch
    dev = torch.device("cuda", 0)
    c = B.compress_chunk(synth.int_for(4096), None, "for(bp)")
    src = torch.zeros(c.size + 16, dtype=torch.uint8, device=dev)
    src[:c.size] = torch.from_numpy(c)
    out = torch.zeros(4096 * 4, dtype=torch.uint8, device=dev)
    d = N.ChunkDesc()
    d.host_src, d.dev_src, d.src_bytes, d.dev_dst, d.dst_bytes = c.ctypes.data, src.data_ptr(), c.size, out.data_ptr(), 4096 * 4
    too_many = (N.ChunkDesc * 257)(*([d] * 257))
    st = C.c_uint32()
    rc = gpu_ctx.lib.btr_gpu_decode_chunks_noalloc(gpu_ctx.handle, too_many, 257, api.current_stream_ptr(), C.byref(st))
    assert rc == 1  # BTR_ERR_INVALID_ARGUMENT: at most 256 chunks per call
    # a damaged stream is reported, and the next call is clean again
    bad = c.copy()
    bad[40:56] = 0xFF
    bsrc = torch.zeros(bad.size + 16, dtype=torch.uint8, device=dev)
    bsrc[:bad.size] = torch.from_numpy(bad)
    db = N.ChunkDesc()
    db.host_src, db.dev_src, db.src_bytes, db.dev_dst, db.dst_bytes = bad.ctypes.data, bsrc.data_ptr(), bad.size, out.data_ptr(), 4096 * 4
    rc = gpu_ctx.lib.btr_gpu_decode_chunks_noalloc(gpu_ctx.handle, (N.ChunkDesc * 1)(db), 1, api.current_stream_ptr(), C.byref(st))
    assert rc in (0, 2)
    rc = gpu_ctx.lib.btr_gpu_decode_chunks_noalloc(gpu_ctx.handle, (N.ChunkDesc * 1)(d), 1, api.current_stream_ptr(), C.byref(st))
    assert rc == 0 and st.value == 0
    assert np.array_equal(out.cpu().numpy().view(np.int32), synth.int_for(4096))
