"""GPU: the BtrReader / Datablock mirrors, device-resident whole-column decode and malformed
stream detection, all through the C ABI."""
import os

import numpy as np
import pytest

import btrblocks_amd as B
from btrblocks_amd import synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _bits(a):
    return a.view(np.uint32 if a.dtype == np.int32 else np.uint64)


@pytest.mark.parametrize("gen,spec", [("int_rle", "rle(bp,bp)"), ("int_dict", "dict(bp)"), ("int_for", "for(bp)"),
                                      ("double_decimal", "pseudodecimal(bp,bp,uncompressed)"),
                                      ("int_dict", "auto"), ("double_decimal", "auto"),
                                      # tools/examples/compression.cpp:10-20 at 1 000 000 rows: mt19937(42) % 4095 in
                                      # runs of 40, default scheme set
                                      ("example_runs", "auto")])
def test_reader_decodes_whole_column(gpu_ctx, gen, spec):
    rows = 1_000_000  # 16 blocks: 15 full + 16960 rows
    values = getattr(synth, gen)(rows)
    image = B.compress_column(values, None, spec, threads=4)
    r = B.BtrReader(image, ctx=gpu_ctx)
    assert r.getChunkCount() == 16 and r.getTupleCount(15) == 16960
    r.decodeAll()
    got = r.column_host()
    np.testing.assert_array_equal(_bits(got), _bits(values))
    one = r.readColumn(3)
    np.testing.assert_array_equal(_bits(one), _bits(values[3 * 65536:4 * 65536]))


def test_reader_with_nullmaps(gpu_ctx):
    rows = 300_000
    values = synth.int_dict(rows)
    nm = synth.random_nullmap(rows, 0.8)
    nm[65536:131072] = 1       # ALLONES block
    nm[131072:196608] = 0      # ALLZEROS block
    image = B.compress_column(values, nm, "auto", threads=2)
    r = B.BtrReader(image, ctx=gpu_ctx)
    r.decodeAll(with_nullmap=True)
    got = r.column_host()
    got_nm = r.d_nullmap[:rows].cpu().numpy()
    np.testing.assert_array_equal(got_nm, nm)
    valid = nm.astype(bool)
    np.testing.assert_array_equal(got[valid], values[valid])
    for i, (o, n) in enumerate(r.chunks):  # every row, null or not, equals the CPU restatement
        want = pyoracle.decode_chunk(image[o:o + n])
        np.testing.assert_array_equal(got[i * 65536:i * 65536 + want.size], want)
    np.testing.assert_array_equal(r.getBitmap(0), nm[:65536])


@pytest.mark.parametrize("double", [False, True])
@pytest.mark.parametrize("nulls", [None, 0.9])
def test_reader_column_of_mixed_chunks(gpu_ctx, double, nulls):
    """Every block of the column has another character, so the compressor picks another cascade per
    block and one decode call launches several kernel kinds side by side."""
    n = 65536
    if double:
        parts = [synth.double_decimal(n), synth.double_dict(n), synth.double_runs(n), synth.double_decimal_runs(n),
                 np.full(n, 2.5), np.random.default_rng(5).random(n), synth.double_dict(20000)]
    else:
        parts = [synth.int_rle(n), synth.int_dict(n), synth.int_for(n), synth.int_outliers(n),
                 np.full(n, 7, np.int32), synth.int_frequency(n), synth.int_dict(20000)]
    values = np.concatenate(parts)
    nm = None if nulls is None else synth.random_nullmap(values.size, nulls, seed=9)
    image = B.compress_column(values, nm, "auto", threads=4)
    r = B.BtrReader(image, ctx=gpu_ctx)
    assert r.getChunkCount() == len(parts)
    r.decodeAll(with_nullmap=nm is not None)
    got = r.column_host()
    for i, (o, c) in enumerate(r.chunks):
        want = pyoracle.decode_chunk(image[o:o + c])
        np.testing.assert_array_equal(_bits(got[i * n:i * n + want.size]), _bits(want))
    if nm is None:
        np.testing.assert_array_equal(_bits(got), _bits(values))
    else:
        np.testing.assert_array_equal(r.d_nullmap[:values.size].cpu().numpy(), nm)
        valid = nm.astype(bool)
        np.testing.assert_array_equal(_bits(got[valid]), _bits(values[valid]))


def test_datablock_decompress(gpu_ctx):
    n = 640000  # tools/examples/compression.cpp compresses the whole relation as one block
    cols = [(synth.example_runs(n, seed=42), None, "auto"),
            (synth.example_runs(n, seed=69, dtype=np.float64), None, "auto"),
            (synth.int_for(n), synth.random_nullmap(n, 0.5), "for(bp)")]
    block = B.compress_datablock(cols)
    want = pyoracle.decode_datablock(block)
    got = B.Datablock(ctx=gpu_ctx).decompress(block)
    for (values, nm, _), (w, w_nm), (g, g_nm) in zip(cols, want, got):
        np.testing.assert_array_equal(_bits(g), _bits(w))
        np.testing.assert_array_equal(g_nm, w_nm)
        valid = np.ones(n, bool) if nm is None else nm.astype(bool)
        np.testing.assert_array_equal(g[valid], values[valid])


def test_malformed_streams_are_reported_not_crashed(gpu_ctx):
    chunk = B.compress_chunk(synth.int_for(65536), None, "for(bp)")
    rng = np.random.default_rng(3)
    for trial in range(8):
        bad = chunk.copy()
        lo = 40 + int(rng.integers(0, bad.size - 100))
        bad[lo:lo + 16] = 0xFF  # garbage inside the bit-packed payload (may hit a width header)
        try:
            out = gpu_ctx.decode_chunk_host(bad)
            assert out.shape == (65536,)
        except B.BtrError as e:
            assert e.status == 2  # BTR_ERR_CORRUPT
    # a run table that does not add up
    c = B.compress_chunk(synth.int_rle(65536), None, "rle(uncompressed,uncompressed)")
    bad = c.copy()
    runs = int(bad[12:16].view(np.uint32)[0])
    counts_off = int(bad[16:20].view(np.uint32)[0])
    first_count = 12 + 10 + counts_off
    bad[first_count:first_count + 4] = np.array([70000], dtype=np.uint32).view(np.uint8)
    assert runs > 10
    with pytest.raises(B.BtrError) as e:
        gpu_ctx.decode_chunk_host(bad)
    assert e.value.status == 2


@pytest.mark.parametrize("gen,spec", [("int_for", "for(bp)"), ("int_rle", "rle(bp,bp)"), ("int_dict", "dict(bp)"),
                                      ("double_decimal", "pseudodecimal(bp,bp,uncompressed)"),
                                      ("int_outliers", "pfor")])
def test_full_size_column_properties(gpu_ctx, gen, spec):
    """Full benchmark size (268 435 456 rows, 4096 blocks): lossless against the generator, checked
    by a checksum of per-block position-weighted checksums computed on the device (64-bit wrapping
    arithmetic over the value bit patterns), plus a stride of blocks bit for bit against the oracle."""
    import torch
    rows = 268_435_456
    values = getattr(synth, gen)(rows)
    image = B.compress_column(values, None, spec)
    r = B.BtrReader(image, ctx=gpu_ctx)
    assert r.getChunkCount() == 4096
    out = r.decodeAll()
    wide = values.dtype == np.float64
    got = out[:values.nbytes].view(torch.int64 if wide else torch.int32).view(4096, 65536)
    weights = torch.arange(1, 65537, dtype=torch.int64, device=got.device)
    sums = torch.empty(4096, dtype=torch.int64)
    for lo in range(0, 4096, 512):  # in slices: the int64 products of a whole column would not fit beside it
        sums[lo:lo + 512] = (got[lo:lo + 512].to(torch.int64) * weights).sum(dim=1).cpu()
    bits = values.view(np.int64 if wide else np.int32).reshape(4096, 65536)
    w = np.arange(1, 65537, dtype=np.int64)
    ref = np.empty(4096, dtype=np.int64)
    with np.errstate(over="ignore"):
        for lo in range(0, 4096, 512):
            ref[lo:lo + 512] = (bits[lo:lo + 512].astype(np.int64) * w).sum(axis=1)
    assert np.array_equal(sums.numpy(), ref)
    del got, out
    for i in sorted(set(range(0, 4096, 409)) | {1234, 4095}):
        o, n = r.chunks[i]
        want = pyoracle.decode_chunk(image[o:o + n])
        np.testing.assert_array_equal(_bits(r.readColumn(i)), _bits(want))
        np.testing.assert_array_equal(_bits(want), _bits(values[i * 65536:(i + 1) * 65536]))


@pytest.mark.gpu
@pytest.mark.parametrize("spec", ["auto", "rle", "dict"])
def test_cpp_interface_round_trip(spec):
    """examples/compression: Relation -> Datablock::compress -> Datablock::decompress through
    include/btrblocks_amd.hpp, ten 65536-row blocks of an integer and a double column."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "compression")
    r = subprocess.run([exe, "640000", spec], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "block round trip: every block decodes to its input" in r.stdout
    assert "part reader round trip: every chunk decodes to its input" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("spec,gen", [
    ("rle(bp,bp)", "int_rle"), ("dict(bp)", "int_dict"), ("pfor", "int_outliers"), ("rle(pfor,pfor)", "int_rle"),
    ("frequency(bp)", "int_frequency"), ("pseudodecimal(bp,bp,uncompressed)", "double_decimal"),
    ("dict(bp)", "double_dict"), ("rle(uncompressed,bp)", "double_runs"),
    ("pseudodecimal", "double_decimal"), ("for(bp)", "int_for"),
    ("pseudodecimal(rle(bp,bp),rle(bp,bp),uncompressed)", "double_decimal_runs"),
])
def test_corrupted_payloads_never_fault(gpu_ctx, spec, gen):
    """Random damage anywhere behind the chunk header: the decode either completes (with whatever
    values the damaged stream describes) or reports BTR_ERR_CORRUPT / BTR_ERR_UNSUPPORTED; it never
    reads or writes outside its buffers."""
    values = getattr(synth, gen)(65536)
    chunk = B.compress_chunk(values, None, spec)
    import zlib
    rng = np.random.default_rng(zlib.crc32((spec + gen).encode()))
    for trial in range(6):
        bad = chunk.copy()
        for _ in range(int(rng.integers(1, 4))):
            lo = 12 + int(rng.integers(0, bad.size - 12 - 8))
            bad[lo:lo + 8] = rng.integers(0, 256, 8).astype(np.uint8)
        try:
            out = gpu_ctx.decode_chunk_host(bad)
            assert out.shape == (65536,)
        except B.BtrError as e:
            assert e.status in (2, 3, 6), e
    # the undamaged chunk still decodes bit-exactly afterwards
    out = gpu_ctx.decode_chunk_host(chunk)
    assert np.array_equal(np.asarray(out).view(np.uint8), values.view(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("spec,gen", [
    ("dict(bp)", "int_dict"), ("for(bp)", "int_for"), ("pseudodecimal", "double_decimal"), ("dict(bp)", "double_dict"),
    ("rle(bp,bp)", "int_rle"),
])
def test_corrupted_payloads_with_nullmap_never_fault(gpu_ctx, spec, gen):
    """The same with a Roaring nullmap in the chunk and the byte-per-row map requested (the damage
    may hit the serialized bitmap as well)."""
    values = getattr(synth, gen)(65536)
    nullmap = synth.random_nullmap(65536, 0.8, seed=21)
    chunk = B.compress_chunk(values, nullmap, spec)
    import zlib
    rng = np.random.default_rng(zlib.crc32((spec + gen + "nm").encode()))
    for trial in range(6):
        bad = chunk.copy()
        for _ in range(int(rng.integers(1, 4))):
            lo = 12 + int(rng.integers(0, bad.size - 12 - 8))
            bad[lo:lo + 8] = rng.integers(0, 256, 8).astype(np.uint8)
        try:
            out, nm = gpu_ctx.decode_chunk_host(bad, want_nullmap=True)
            assert out.shape == (65536,) and nm.shape == (65536,)
        except B.BtrError as e:
            assert e.status in (2, 3, 6), e
    out, nm = gpu_ctx.decode_chunk_host(chunk, want_nullmap=True)
    assert np.array_equal(nm.astype(bool), nullmap.astype(bool))


@pytest.mark.gpu
def test_relation_directory_round_trip(gpu_ctx, tmp_path):
    """Write a relation directory in the reference's layout, open it, decode every column on the GPU."""
    from btrblocks_amd import files
    rows = 200_000
    cols = [synth.int_rle(rows), synth.double_decimal(rows), synth.int_for(rows)]
    files.write_directory(str(tmp_path), cols, scheme=["rle(bp,bp)", "pseudodecimal(bp,bp,uncompressed)", "auto"])
    d = files.BtrDirectory(str(tmp_path), ctx=gpu_ctx)
    assert d.num_chunks == 4 and len(d.columns) == 3
    for c, want in enumerate(cols):
        got = d.decode_column(c)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    # every column held at once, compared after the last decode and after the directory object is
    # gone: a returned array owns the pinned memory it views
    held = [d.decode_column(c) for c in range(len(cols))]
    held_nm = [d.decode_column(c, with_nullmap=True)[1] for c in range(len(cols))]
    del d
    import gc
    gc.collect()
    for got, nm, want in zip(held, held_nm, cols):
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
        assert nm.shape == (rows,) and bool(nm.all())


@pytest.mark.gpu
def test_columns_decoded_beside_each_other_match_one_after_the_other(gpu_ctx):
    """btr_gpu_decode_columns (decompression-speed.cpp:45-58 decodes columns in parallel) against
    btr_gpu_decode per plan: same bytes, twice in a row, with more plans than lanes."""
    import torch
    from btrblocks_amd import api
    specs = [("int_rle", "rle(bp,bp)"), ("int_dict", "dict(bp)"), ("int_for", "for(bp)"), ("double_decimal", "pseudodecimal"),
             ("int_outliers", "pfor"), ("double_dict", "dict(bp)"), ("double_runs", "rle(uncompressed,bp)"),
             ("int_frequency_pool", "frequency(dict(bp))"), ("int_constant", "onevalue"), ("int_rle_pool", "rle(dict(bp),bp)")]
    readers = []
    for k, (gen, spec) in enumerate(specs):
        values = getattr(synth, gen)(3 * 65536 + 1000 * k)
        readers.append(B.BtrReader(B.compress_column(values, None, spec), ctx=gpu_ctx))
    plans = [r.plan(None, False) for r in readers]
    stream = api.current_stream_ptr()
    for r in readers:
        r.d_out.zero_()
    for p in plans:
        p.decode(stream)
    torch.cuda.synchronize()
    want = [r.d_out.clone() for r in readers]
    for _ in range(2):
        for r in readers:
            r.d_out.zero_()
        api.decode_columns(gpu_ctx, plans, stream)
        torch.cuda.synchronize()
        for p in plans:
            assert p.check(stream) == 0
        for r, w in zip(readers, want):
            assert torch.equal(r.d_out, w)


@pytest.mark.parametrize("nulls", [None, 0.9])
def test_relation_level_plan_over_all_columns(gpu_ctx, nulls):
    """One plan over the chunks of every column of a relation (decompression-speed.cpp:45-58 hands all
    columns x parts x chunks to one scheduler): its launches are kernels and occupancy classes, not
    columns; the FOR, DICT and Pseudodecimal chunks share one mixed launch. Decoded twice back to
    back; every chunk bit for bit against the oracle."""
    import torch
    from btrblocks_amd import api, _native as N
    rows = 5 * 65536 + 1234  # ragged last block
    cols = [("int_rle", "rle(bp,bp)"), ("int_dict", "dict(bp)"), ("int_for", "for(bp)"), ("int_rle_pool", "rle(dict(bp),bp)"),
            ("int_outliers", "pfor"), ("int_constant", "onevalue"), ("int_frequency_pool", "frequency(dict(bp))"),
            ("double_decimal", "pseudodecimal(bp,bp,uncompressed)"),
            ("double_decimal_runs", "pseudodecimal(rle(dict(bp),bp),rle(bp,bp),uncompressed)"),
            ("double_dict", "dict(bp)"), ("double_runs", "rle(uncompressed,bp)")]
    readers = []
    for k, (gen, spec) in enumerate(cols):
        values = getattr(synth, gen)(rows)
        nm = None if nulls is None else synth.random_nullmap(rows, nulls, seed=20 + k)
        readers.append(B.BtrReader(B.compress_column(values, nm, spec, threads=2), ctx=gpu_ctx))
    want_nm = nulls is not None
    for r in readers:
        r.plan(None, want_nm)  # allocates the nullmap outputs
    descs = [r.chunk_desc(i, want_nm) for r in readers for i in range(len(r.chunks))]
    plan = api.DecodePlan(gpu_ctx, descs, keepalive=tuple(readers))
    st = plan.stats
    assert st.chunks == len(descs) and st.launches >= 4 and st.launches < len(cols)
    mixed_bit = 16 + (1 if want_nm else 0)
    assert st.kernel_mask >> mixed_bit & 1, "FOR, DICT and Pseudodecimal chunks share the mixed stream launch"
    stream = api.current_stream_ptr()
    # two more plans, each over half of the columns: decoded beside each other below (plans with
    # several large launches on the lanes btr_gpu_decode_columns itself decodes on)
    half = len(readers) // 2
    halves = [api.DecodePlan(gpu_ctx, [r.chunk_desc(i, want_nm) for r in part for i in range(len(r.chunks))], keepalive=tuple(part))
              for part in (readers[:half], readers[half:])]
    assert all(h.stats.launches > 1 for h in halves)
    for attempt in range(3):
        for r in readers:
            r.d_out.zero_()
        if attempt < 2:
            plan.decode(stream)
            plan.check(stream)
        else:
            api.decode_columns(gpu_ctx, halves, stream)
            for h in halves:
                h.check(stream)
        torch.cuda.synchronize()
        for r in readers:
            got = r.d_out.cpu().numpy()
            got_nm = r.d_nullmap.cpu().numpy() if want_nm else None
            for i, (o, n) in enumerate(r.chunks):
                chunk = r.image[o:o + n]
                want, oracle_nm = pyoracle.decode_chunk(chunk, want_nullmap=True)
                want = want.view(np.uint8)
                lo = int(r.out_offsets[i])
                np.testing.assert_array_equal(got[lo:lo + want.size], want)
                if want_nm:
                    r0 = int(r.row_offsets[i])
                    np.testing.assert_array_equal(got_nm[r0:r0 + oracle_nm.size], oracle_nm)
    plan.close()
    for h in halves:
        h.close()
