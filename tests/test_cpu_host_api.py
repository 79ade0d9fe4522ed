"""CPU suite: the C ABI library loads and exports every declared symbol, host-side inspection and
the plan compiler's validation (no GPU compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

import btrblocks_amd as B
from btrblocks_amd import _native, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "btrblocks_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(btr_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/btrblocks_amd.h but not exported"
    assert declared == set(_native.SIGNATURES), "ctypes binding and header disagree"


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(B.BtrError) as e:
        B.GpuContext(0)
    assert e.value.status == 5  # BTR_ERR_DEVICE: no CPU fallback for decoding


def test_describe_and_inspect():
    c = B.compress_chunk(synth.example_runs(65536, unique=50, runlength=7), None, "rle(dict(bp),bp)")
    assert B.describe(c) == "RLE(DICT(BP),BP)"
    info = B.chunk_info(c)
    assert (info.tuple_count, info.column_type, info.scheme, info.nullmap_type) == (65536, 0, 3, 0)
    assert info.decoded_bytes == 65536 * 4
    c = B.compress_chunk(synth.double_decimal(1000), synth.random_nullmap(1000, 0.3), "pseudodecimal(bp,bp,uncompressed)")
    assert B.describe(c) == "PSEUDODECIMAL(BP,BP,UNCOMPRESSED)"
    assert B.chunk_info(c).nullmap_type == 2  # REGULAR: fewer valid rows than null rows


def test_plan_compiler_rejects_corrupt_headers():
    c = B.compress_chunk(synth.int_dict(4096, distinct=100), None, "dict(bp)")
    with pytest.raises(B.BtrError):
        B.describe(c[:40])  # dictionary runs past the end
    bad = c.copy()
    bad[0] = 77  # unknown scheme code
    with pytest.raises(B.BtrError):
        B.describe(bad)
    bad = c.copy()
    bad[2] = 2  # string column
    with pytest.raises(B.BtrError):
        B.chunk_info(bad)
    with pytest.raises(B.BtrError):
        B.compress_chunk(np.zeros(4, np.int32), None, "pseudodecimal")  # not an integer scheme
    with pytest.raises(B.BtrError):
        B.compress_chunk(np.zeros(4, np.int32), None, "rle(bp,bp,bp)")


def test_plan_compiler_rejects_absurd_row_counts():
    """A damaged tuple_count near 2^30 must not wrap the 32-bit table sizes of the plan: it is
    refused before any table is placed."""
    c = B.compress_chunk(synth.int_rle(4096), None, "rle(bp,bp)")
    for rows in (2**30 - 1, 2**30 + 5, 2**32 - 1, 2**24 + 1):
        bad = c.copy()
        bad[8:12] = np.array([rows], dtype=np.uint32).view(np.uint8)
        with pytest.raises(B.BtrError) as e:
            B.describe(bad)
        assert e.value.status in (2, 3)
    c = B.compress_chunk(synth.int_for(4096), None, "pfor")
    bad = c.copy()
    bad[8:12] = np.array([2**30], dtype=np.uint32).view(np.uint8)
    with pytest.raises(B.BtrError):
        B.describe(bad)


def test_part_inspect_matches_per_chunk_calls():
    values = synth.int_for(200000)
    image = B.compress_column(values, None, "for(bp)", threads=2)
    infos, offsets, lengths = B.api.part_inspect(image)
    assert len(infos) == 4
    for i, (o, n) in enumerate(B.part_chunks(image)):
        one = B.chunk_info(image[o:o + n])
        assert (offsets[i], lengths[i]) == (o, n)
        assert (infos[i].tuple_count, infos[i].scheme, infos[i].decoded_bytes) == (one.tuple_count, one.scheme, one.decoded_bytes)
    with pytest.raises(B.BtrError):
        B.api.part_inspect(image[:30])


def test_column_part_image_layout():
    values = synth.int_for(200000)
    image = B.compress_column(values, None, "for(bp)", threads=2)
    chunks = B.part_chunks(image)
    assert len(chunks) == 4
    assert all(off % 16 == 0 for off, _ in chunks)  # storage/Chunk.cpp:261-272
    rows = [B.chunk_info(image[o:o + n]).tuple_count for o, n in chunks]
    assert rows == [65536, 65536, 65536, 200000 - 3 * 65536]
    assert int(image[:4].view(np.uint32)[0]) == 4


def test_auto_picker_is_deterministic_and_sensible():
    a = B.compress_chunk(synth.int_rle(65536), None, "auto")
    b = B.compress_chunk(synth.int_rle(65536), None, "auto")
    np.testing.assert_array_equal(a, b)
    assert B.describe(a).startswith("RLE(")
    assert B.describe(B.compress_chunk(synth.double_decimal(65536), None, "auto")).startswith("PSEUDODECIMAL(")
    assert B.describe(B.compress_chunk(np.full(100, 3, np.int32), None, "auto")) == "ONE_VALUE"
    rng = np.random.default_rng(0)
    assert B.describe(B.compress_chunk(rng.uniform(-1e6, 1e6, 65536), None, "auto")) == "UNCOMPRESSED"


def test_cpp_example_fails_loudly_without_gpu():
    """The C++ interface example (include/btrblocks_amd.hpp) is built in-tree; without a HIP device
    it must stop with the device error instead of decoding on the CPU."""
    import subprocess
    import torch
    exe = os.path.join(ROOT, "examples", "compression")
    assert os.path.exists(exe), "run python -m btrblocks_amd.build"
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe, "1000"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2
    assert "no HIP device" in r.stderr


def test_relation_directory_layout(tmp_path):
    """metadata + column<c>_part<p> files as the reference's tools write them: every part is a part
    image (num_chunks, offsets, 16-byte aligned chunks), parts respect the 16 MiB limit, and the
    oracle decodes every chunk back to the input."""
    from btrblocks_amd import files
    from oracle import pyoracle
    rows = 300_000
    ints = synth.int_dict(rows)
    dbls = synth.double_decimal(rows)
    num_chunks, parts = files.write_directory(str(tmp_path), [ints, dbls], scheme=["dict(bp)", "auto"])
    assert num_chunks == 5
    got_chunks, cols = files.read_metadata(str(tmp_path))
    assert got_chunks == 5 and [c[0] for c in cols] == [0, 1] and [c[1] for c in cols] == parts
    for c, want in enumerate([ints, dbls]):
        decoded = []
        for p in range(parts[c]):
            image = np.fromfile(str(tmp_path / f"column{c}_part{p}"), dtype=np.uint8)
            assert image.size <= files.PART_BYTES + 4096
            for off, n in B.part_chunks(image):
                assert off % 16 == 0
                decoded.append(pyoracle.decode_chunk(image[off:off + n]))
        got = np.concatenate(decoded)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_relation_directory_splits_parts(tmp_path, monkeypatch):
    from btrblocks_amd import files
    monkeypatch.setattr(files, "PART_BYTES", 300_000)
    vals = np.random.default_rng(5).integers(0, 2**31 - 1, 400_000).astype(np.int32)
    _, parts = files.write_directory(str(tmp_path), [vals], scheme="uncompressed")
    assert parts[0] > 1
    total = sum(len(B.part_chunks(np.fromfile(str(tmp_path / f"column0_part{p}"), dtype=np.uint8))) for p in range(parts[0]))
    assert total == 7
