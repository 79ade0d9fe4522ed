"""GPU: the integer sub-streams of string DICT chunks (codes, FSST lengths, the RLE-optimised
path's run values / run lengths) decoded by the integer kernels through the C ABI, bit-exact
against the CPU restatement."""
import numpy as np
import pytest
import torch

import btrblocks_amd as B
from btrblocks_amd import _native as N
from btrblocks_amd import api
from oracle import pyoracle
from strings_cases import cases

pytestmark = pytest.mark.gpu
CASES = cases()
GUARD = 64


def _streams(info):
    which = [(N.SUBSTREAM_CODES, info.tuple_count)]
    if info.use_fsst:
        which.append((N.SUBSTREAM_LENGTHS, info.num_codes))
    if info.use_rle_optimized_path:
        which += [(N.SUBSTREAM_RUN_VALUES, info.runs_count), (N.SUBSTREAM_RUN_LENGTHS, info.runs_count)]
    return which


def _decode_all(gpu_ctx, chunks):
    """One plan over every sub-stream of every chunk; returns [(chunk index, which, values)]."""
    dev = torch.device("cuda", 0)
    keep, descs, outs = [], [], []
    for ci, chunk in enumerate(chunks):
        d_src = torch.zeros(chunk.size + 16, dtype=torch.uint8, device=dev)
        d_src[:chunk.size] = torch.from_numpy(chunk)
        keep.append(d_src)
        for which, count in _streams(api.string_dict_info(chunk)):
            d_out = torch.full((GUARD + count * 4 + GUARD,), 0xA5, dtype=torch.uint8, device=dev)
            d = N.SubstreamDesc()
            d.host_src = chunk.ctypes.data
            d.dev_src = d_src.data_ptr()
            d.src_bytes = chunk.size
            d.which = which
            d.dev_dst = d_out.data_ptr() + GUARD
            d.dst_bytes = count * 4
            descs.append(d)
            outs.append((ci, which, count, d_out))
    plan = B.DecodePlan.for_string_dict(gpu_ctx, descs, keepalive=(keep, chunks))
    plan.decode(api.current_stream_ptr())
    assert plan.check(api.current_stream_ptr()) == 0
    res = []
    for ci, which, count, d_out in outs:
        host = d_out.cpu().numpy()
        assert (host[:GUARD] == 0xA5).all() and (host[GUARD + count * 4:] == 0xA5).all()
        res.append((ci, which, host[GUARD:GUARD + count * 4].copy().view(np.int32)))
    plan.close()
    return res


def test_every_sub_stream_matches_the_oracle(gpu_ctx):
    chunks = [api.compress_string_dict_chunk(s, None, cs, ls, fsst) for _, s, cs, ls, fsst in CASES]
    seen = set()
    for ci, which, got in _decode_all(gpu_ctx, chunks):
        want = pyoracle.string_dict_stream(chunks[ci], which)
        np.testing.assert_array_equal(got, want, err_msg=f"{CASES[ci][0]} stream {which}")
        seen.add(which)
    assert seen == {0, 1, 2, 3}


def test_codes_rebuild_the_rows(gpu_ctx):
    """Codes decoded on the GPU, looked up in the chunk's dictionary, give back the input rows."""
    name, strings, cs, ls, fsst = CASES[0]
    chunk = api.compress_string_dict_chunk(strings, None, cs, ls, fsst)
    info = api.string_dict_info(chunk)
    (_, _, codes), = [r for r in _decode_all(gpu_ctx, [chunk]) if r[1] == N.SUBSTREAM_CODES]
    slots = chunk[info.pool_offset:info.pool_offset + 4 * (info.num_codes + 1)].copy().view(np.uint32)
    pool = chunk[info.pool_offset:]
    rows = [pool[slots[c]:slots[c + 1]].tobytes() for c in codes]
    assert rows == list(strings)


def test_wrong_requests_are_refused(gpu_ctx):
    chunk = api.compress_string_dict_chunk([b"a", b"b", b"c", b"a"] * 100, None, "bp")
    dev = torch.device("cuda", 0)
    d_src = torch.zeros(chunk.size + 16, dtype=torch.uint8, device=dev)
    d_src[:chunk.size] = torch.from_numpy(chunk)
    d_out = torch.zeros(4096, dtype=torch.uint8, device=dev)

    def desc(which, dst_bytes=1600, src=chunk):
        d = N.SubstreamDesc()
        d.host_src, d.dev_src, d.src_bytes = src.ctypes.data, d_src.data_ptr(), src.size
        d.which, d.dev_dst, d.dst_bytes = which, d_out.data_ptr(), dst_bytes
        return d

    for which, status in [(N.SUBSTREAM_LENGTHS, 1), (N.SUBSTREAM_RUN_VALUES, 1), (7, 1)]:
        with pytest.raises(B.BtrError) as e:
            B.DecodePlan.for_string_dict(gpu_ctx, [desc(which)])
        assert e.value.status == status
    with pytest.raises(B.BtrError) as e:
        B.DecodePlan.for_string_dict(gpu_ctx, [desc(N.SUBSTREAM_CODES, dst_bytes=100)])
    assert e.value.status == 4
    ints = B.compress_chunk(np.arange(100, dtype=np.int32))
    with pytest.raises(B.BtrError) as e:
        B.DecodePlan.for_string_dict(gpu_ctx, [desc(N.SUBSTREAM_CODES, src=ints)])
    assert e.value.status == 3
