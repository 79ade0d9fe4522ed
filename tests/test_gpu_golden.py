"""GPU: the committed golden chunk bytes (tests/golden/*.npz, inputs from
tools/playground/for_tests.cpp:47-58, tools/examples/compression.cpp:10-20 and the
test/DatasetGenerator.cpp distributions) decoded by the HIP path through the C ABI and compared
with the stored values and with the CPU oracle. The stored bytes are decoded as they are, not
re-compressed: a drift between a committed chunk and today's kernels shows up here."""
import glob
import os

import numpy as np
import pytest

import btrblocks_amd as B
from oracle import pyoracle

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _bits(a):
    return a.view(np.uint32 if a.dtype == np.int32 else np.uint64)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_gpu_decodes_committed_chunk(gpu_ctx, path):
    f = np.load(path, allow_pickle=False)
    chunk, values, nullmap = np.ascontiguousarray(f["chunk"]), f["values"], f["nullmap"]
    want, want_nm = pyoracle.decode_chunk(chunk, want_nullmap=True)
    got, got_nm = gpu_ctx.decode_chunk_host(chunk, want_nullmap=True)
    plain = gpu_ctx.decode_chunk_host(chunk)
    assert got.dtype == values.dtype and got.shape == values.shape
    # every row, null or not, as the oracle decodes the same bytes
    np.testing.assert_array_equal(_bits(got), _bits(want))
    np.testing.assert_array_equal(_bits(np.asarray(plain)), _bits(want))
    np.testing.assert_array_equal(got_nm, want_nm)
    # the stored expected output
    np.testing.assert_array_equal(got_nm, nullmap)
    valid = nullmap.astype(bool)
    np.testing.assert_array_equal(got[valid], values[valid])
    if "pseudo" in os.path.basename(path) or "doublebp" in os.path.basename(path):
        np.testing.assert_array_equal(_bits(got), _bits(values))  # -0.0, denormals, huge values bit for bit
