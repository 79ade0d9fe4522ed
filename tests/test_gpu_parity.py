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
    if values.dtype == np.float64 and ("rle" in spec or "dict" in spec or "frequency" in spec or "auto" in spec
                                       or "onevalue" in spec):
        # value-equality based schemes merge -0.0 with 0.0, like the reference; everything else
        # (NaN payloads included) keeps its bits
        zero = valid & (values == 0)
        assert np.all(got[zero] == 0)
        np.testing.assert_array_equal(_bits(got)[valid & ~zero], _bits(values)[valid & ~zero])
    else:
        np.testing.assert_array_equal(_bits(got)[valid], _bits(values)[valid])


@pytest.mark.parametrize("offset", [0, 8])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_nothing_is_written_outside_the_column(gpu_ctx, case, offset):
    """The reference over-allocates every output by SIMD_EXTRA_BYTES because its loops overshoot
    (common/SIMD.hpp:30-31); a C-ABI caller promises no slack. Guard bytes in front of and behind
    the caller-owned column and nullmap buffers stay untouched, with the column 16-byte aligned
    and only 8-byte aligned."""
    import torch
    from btrblocks_amd import _native as N
    from btrblocks_amd import api
    name, values, nullmap, spec, kw = case
    chunk = B.compress_chunk(values, nullmap, spec, **kw)
    want, want_nm = pyoracle.decode_chunk(chunk, want_nullmap=True)
    n, nbytes = values.size, values.nbytes
    guard = 64
    dev = torch.device("cuda", 0)
    d_src = torch.zeros(chunk.size + 16, dtype=torch.uint8, device=dev)
    d_src[:chunk.size] = torch.from_numpy(chunk)
    d_out = torch.full((guard + offset + nbytes + guard,), 0xA5, dtype=torch.uint8, device=dev)
    d_nm = torch.full((guard + n + guard,), 0xA5, dtype=torch.uint8, device=dev)
    d = N.ChunkDesc()
    d.host_src = chunk.ctypes.data
    d.dev_src = d_src.data_ptr()
    d.src_bytes = chunk.size
    d.dev_dst = d_out.data_ptr() + guard + offset
    d.dst_bytes = nbytes
    d.dev_nullmap_dst = d_nm.data_ptr() + guard
    assert d.dev_dst % 8 == 0
    plan = api.DecodePlan(gpu_ctx, [d], keepalive=(d_src, d_out, d_nm, chunk))
    s = api.current_stream_ptr()
    plan.decode(s)
    plan.check(s)
    out = d_out.cpu().numpy()
    nm = d_nm.cpu().numpy()
    plan.close()
    lo = guard + offset
    assert np.all(out[:lo] == 0xA5) and np.all(out[lo + nbytes:] == 0xA5), f"{name}: wrote outside the column"
    assert np.all(nm[:guard] == 0xA5) and np.all(nm[guard + n:] == 0xA5), f"{name}: wrote outside the nullmap"
    np.testing.assert_array_equal(out[lo:lo + nbytes], want.view(np.uint8))
    np.testing.assert_array_equal(nm[guard:guard + n], want_nm)
