"""Seeded random sweep: sizes, trees and nullmap densities that the hand-written case list does not
enumerate, GPU against the oracle at every row, with and without the byte-per-row nullmap."""
import os

import numpy as np
import pytest

import btrblocks_amd as B
from btrblocks_amd import synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu

INT_SPECS = ["bp", "for(bp)", "dict(bp)", "rle(bp,bp)", "rle(dict(bp),bp)", "frequency(bp)", "pfor", "for(pfor)", "auto",
             "frequency(dict(bp))", "dict(pfor)", "rle(pfor,pfor)", "pfor_delta", "for(rle(bp,bp))", "dict(rle(bp,bp))",
             "for(dict(bp))", "uncompressed", "dict8", "dict16"]
DBL_SPECS = ["pseudodecimal(bp,bp,uncompressed)", "pseudodecimal", "pseudodecimal(bp,dict(bp),uncompressed)",
             "pseudodecimal(rle(bp,bp),rle(bp,bp),uncompressed)", "dict(bp)", "rle(uncompressed,bp)", "auto",
             "pseudodecimal(pfor,bp,uncompressed)", "pseudodecimal(bp,rle(bp,bp),dict(bp))", "rle(dict(bp),bp)",
             "frequency(uncompressed)", "frequency(pseudodecimal(bp,bp,uncompressed))", "doublebp", "uncompressed",
             "rle(pseudodecimal(bp,bp,uncompressed),bp)"]


def _int_values(rng, n, kind):
    if kind == 0:
        return rng.integers(0, 1 << int(rng.integers(1, 31)), n).astype(np.int32)
    if kind == 1:
        return np.repeat(rng.integers(0, 5000, n // 7 + 1), 7)[:n].astype(np.int32)
    pool = rng.integers(-2**31, 2**31, int(rng.integers(1, 3000))).astype(np.int32)
    return pool[rng.integers(0, pool.size, n)]


def _dbl_values(rng, n, kind):
    if kind == 0:
        v = rng.integers(0, 10**int(rng.integers(1, 8)), n).astype(np.float64) * 10.0 ** -int(rng.integers(0, 5))
    elif kind == 1:
        v = np.repeat(rng.integers(0, 10**6, n // 9 + 1).astype(np.float64) * 0.01, 9)[:n].copy()
    else:
        pool = rng.uniform(-1e6, 1e6, int(rng.integers(1, 3000)))
        v = pool[rng.integers(0, pool.size, n)]
    bad = rng.random(n) < rng.choice([0.0, 0.01, 0.3])
    v[bad] = rng.uniform(-1, 1, int(bad.sum())) / 3.0
    return v


# BTR_SWEEP_FIRST / BTR_SWEEP_CASES: other and larger seed ranges for one-off runs
_FIRST = int(os.environ.get("BTR_SWEEP_FIRST", "0"))


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + int(os.environ.get("BTR_SWEEP_CASES", "300"))))
def test_random_chunk(gpu_ctx, seed):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([1, 17, 127, 128, 129, 1000, 4097, 16960, 40000, 65535, 65536]))
    if rng.random() < 0.4:
        n = int(rng.integers(1, 65537))
    is_double = bool(rng.integers(0, 2))
    kind = int(rng.integers(0, 3))
    values = _dbl_values(rng, n, kind) if is_double else _int_values(rng, n, kind)
    spec = str(rng.choice(DBL_SPECS if is_double else INT_SPECS))
    density = rng.choice([-1.0, 0.0, 0.03, 0.5, 0.97, 1.0])
    nullmap = None if density < 0 else (rng.random(n) < density).astype(np.uint8)
    try:
        chunk = B.compress_chunk(values, nullmap, spec)
    except B.BtrError:
        pytest.skip("the compressor rejects this tree for this data")
    want, want_nm = pyoracle.decode_chunk(chunk, want_nullmap=True)
    view = np.uint64 if is_double else np.uint32
    plain = np.asarray(gpu_ctx.decode_chunk_host(chunk))
    np.testing.assert_array_equal(plain.view(view), want.view(view), err_msg=f"{spec} n={n}: {B.describe(chunk)}")
    got, got_nm = gpu_ctx.decode_chunk_host(chunk, want_nullmap=True)
    np.testing.assert_array_equal(got.view(view), want.view(view), err_msg=f"{spec} n={n}: {B.describe(chunk)}")
    np.testing.assert_array_equal(got_nm, want_nm)
    valid = np.ones(n, bool) if nullmap is None else nullmap.astype(bool)
    np.testing.assert_array_equal(got_nm.astype(bool), valid)
