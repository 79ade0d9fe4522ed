"""GPU parity: the HIP decode path (through the C ABI) against the CPU oracle, bit-exact at every
row, null or not, plus losslessness against the original column at valid rows."""
import numpy as np
import pytest

import btrblocks_amd as B
from oracle import pyoracle

from cases import cases

pytestmark = pytest.mark.gpu

CASES = cases()


def _bits(a):
    return a.view(np.uint32 if a.dtype == np.int32 else np.uint64)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_chunk_matches_oracle(gpu_ctx, case):
    name, values, nullmap, spec, kw = case
    chunk = B.compress_chunk(values, nullmap, spec, **kw)
    want, want_nm = pyoracle.decode_chunk(chunk, want_nullmap=True)
    # without the byte-per-row nullmap (the plan may pick another kernel for the same chunk) ...
    plain = gpu_ctx.decode_chunk_host(chunk)
    np.testing.assert_array_equal(_bits(np.asarray(plain)), _bits(want), err_msg=f"{name}: {B.describe(chunk)}")
    # ... and with it
    got, got_nm = gpu_ctx.decode_chunk_host(chunk, want_nullmap=True)
    assert got.dtype == values.dtype and got.shape == values.shape
    np.testing.assert_array_equal(_bits(got), _bits(want), err_msg=f"{name}: {B.describe(chunk)}")
    np.testing.assert_array_equal(got_nm, want_nm)
    valid = np.ones(values.size, bool) if nullmap is None else nullmap.astype(bool)
    np.testing.assert_array_equal(got_nm.astype(bool), valid)
    if values.dtype == np.float64 and "rle" in spec or "dict" in spec or "frequency" in spec or "auto" in spec:
        # value-equality based schemes merge -0.0 with 0.0, like the reference
        np.testing.assert_array_equal(got[valid], values[valid])
    else:
        np.testing.assert_array_equal(_bits(got)[valid], _bits(values)[valid])
