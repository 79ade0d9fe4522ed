"""CPU suite: the oracle against the committed golden fixtures and against every parity case
(losslessness at valid rows), i.e. the host compressor and the CPU restatement agree."""
import glob
import os

import numpy as np
import pytest

import btrblocks_amd as B
from oracle import pyoracle

from cases import cases

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
CASES = cases()


def _bits(a):
    return a.view(np.uint32 if a.dtype == np.int32 else np.uint64)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_decodes_golden_fixture(path):
    f = np.load(path, allow_pickle=False)
    chunk, values, nullmap = f["chunk"], f["values"], f["nullmap"]
    assert B.describe(chunk) == str(f["tree"])
    got, got_nm = pyoracle.decode_chunk(chunk, want_nullmap=True)
    valid = nullmap.astype(bool)
    np.testing.assert_array_equal(got_nm, nullmap)
    np.testing.assert_array_equal(got[valid], values[valid])
    if "pseudo" in os.path.basename(path) or "doublebp" in os.path.basename(path):
        np.testing.assert_array_equal(_bits(got), _bits(values))  # -0.0, denormals, huge values bit-exact


def test_golden_fixtures_are_current():
    """The committed chunk bytes are what the compressor emits today (format stability)."""
    for path in GOLDEN:
        f = np.load(path, allow_pickle=False)
        nm = None if f["nullmap"].all() and "nulls" not in path else f["nullmap"]
        again = B.compress_chunk(f["values"], nm, str(f["spec"]))
        np.testing.assert_array_equal(again, f["chunk"], err_msg=path)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_compress_then_oracle_is_lossless(case):
    name, values, nullmap, spec, kw = case
    chunk = B.compress_chunk(values, nullmap, spec, **kw)
    info = B.chunk_info(chunk)
    assert info.tuple_count == values.size
    got, got_nm = pyoracle.decode_chunk(chunk, want_nullmap=True)
    valid = np.ones(values.size, bool) if nullmap is None else nullmap.astype(bool)
    np.testing.assert_array_equal(got_nm.astype(bool), valid)
    # bit-exact (NaN payloads and infinities included); value-equality based schemes may merge -0.0
    # with 0.0, like the reference
    zero = valid & (values == 0)
    assert np.all(got[zero] == 0)
    np.testing.assert_array_equal(_bits(got)[valid & ~zero], _bits(values)[valid & ~zero])
    assert chunk.size <= values.nbytes + 64 + (0 if nullmap is None else 4 * values.size + 16)


def test_oracle_rejects_truncated_chunks():
    chunk = B.compress_chunk(np.arange(5000, dtype=np.int32) % 77, None, "dict(bp)")
    for cut in (5, 11, 40, chunk.size // 2, chunk.size - 5):
        with pytest.raises(pyoracle.OracleError):
            pyoracle.decode_chunk(chunk[:cut])


def test_datablock_roundtrip_oracle():
    n = 20000
    from btrblocks_amd import synth
    cols = [(synth.int_rle(n), None, "auto"), (synth.double_decimal(n), synth.random_nullmap(n, 0.5), "auto"),
            (synth.int_dict(n), synth.random_nullmap(n, 0.95), "dict(bp)")]
    block = B.compress_datablock(cols)
    rows, infos = B.datablock_columns(block)
    assert rows == n and len(infos) == 3
    out = pyoracle.decode_datablock(block)
    for (values, nm, _), (got, got_nm) in zip(cols, out):
        valid = np.ones(n, bool) if nm is None else nm.astype(bool)
        np.testing.assert_array_equal(got_nm.astype(bool), valid)
        np.testing.assert_array_equal(got[valid], values[valid])
