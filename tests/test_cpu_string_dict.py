"""CPU: string DICT chunks (scheme/string/DynamicDictionary.cpp) - the host producer, the header
inspection and the oracle's restatement of the integer sub-streams agree with the input strings."""
import numpy as np
import pytest

import btrblocks_amd as B
from btrblocks_amd import api
from oracle import pyoracle
from strings_cases import cases

CASES = cases()


def _expected(strings):
    distinct = sorted(set(strings))
    index = {s: i for i, s in enumerate(distinct)}
    codes = np.array([index[s] for s in strings], dtype=np.int32)
    return distinct, codes


@pytest.mark.parametrize("name,strings,codes_scheme,lengths_scheme,fsst", CASES, ids=[c[0] for c in CASES])
def test_sub_streams_against_the_input(name, strings, codes_scheme, lengths_scheme, fsst):
    chunk = api.compress_string_dict_chunk(strings, None, codes_scheme, lengths_scheme, fsst)
    info = api.string_dict_info(chunk)
    distinct, codes = _expected(strings)
    assert info.tuple_count == len(strings) and info.num_codes == len(distinct)
    assert info.total_decompressed_size == sum(map(len, strings)) + 4 * (len(strings) + 1)
    assert bool(info.use_fsst) == fsst
    np.testing.assert_array_equal(pyoracle.string_dict_stream(chunk, pyoracle.CODES), codes)
    if fsst:
        want = np.array([len(s) for s in distinct], dtype=np.int32)
        np.testing.assert_array_equal(pyoracle.string_dict_stream(chunk, pyoracle.LENGTHS), want)
    else:
        offsets, text = pyoracle.string_dict_rows(chunk)
        assert text.tobytes() == b"".join(strings)
        np.testing.assert_array_equal(np.diff(offsets.astype(np.int64)), [len(s) for s in strings])
    change = np.flatnonzero(np.diff(codes)) + 1
    starts = np.concatenate([[0], change])
    if len(strings) / starts.size > 3.0 and info.codes_scheme == 3:
        assert info.use_rle_optimized_path == 1 and info.runs_count == starts.size
        np.testing.assert_array_equal(pyoracle.string_dict_stream(chunk, pyoracle.RUN_VALUES), codes[starts])
        lengths = np.diff(np.concatenate([starts, [len(strings)]]))
        np.testing.assert_array_equal(pyoracle.string_dict_stream(chunk, pyoracle.RUN_LENGTHS), lengths)
    else:
        assert info.use_rle_optimized_path == 0
        with pytest.raises(pyoracle.OracleError):
            pyoracle.string_dict_stream(chunk, pyoracle.RUN_VALUES)


def test_header_layout_matches_the_reference_struct():
    """DynamicDictionaryStructure is not packed: use_fsst at 4, three padding bytes, data at 27."""
    chunk = api.compress_string_dict_chunk([b"b", b"a", b"b"], None, "uncompressed")
    assert chunk[0] == 2 and chunk[2] == 2                       # DICT scheme, string column
    h = chunk[12:]
    assert h[0:4].view(np.uint32)[0] == 3 + 4 * 4                # total_decompressed_size
    assert h[4] == 0 and h[16:20].view(np.uint32)[0] == 2        # use_fsst, num_codes
    slots = h[27:27 + 12].copy().view(np.uint32)
    assert slots.tolist() == [12, 13, 14] and h[27 + 12:27 + 14].tobytes() == b"ab"
    codes_off = int(h[20:24].copy().view(np.uint32)[0])
    assert codes_off == 14 and h[25] == 0
    assert h[27 + 14:27 + 26].copy().view(np.int32).tolist() == [1, 0, 1]


def test_inspect_rejects_other_chunks_and_bad_offsets():
    with pytest.raises(B.BtrError) as e:
        api.string_dict_info(B.compress_chunk(np.arange(10, dtype=np.int32)))
    assert e.value.status == 3
    chunk = api.compress_string_dict_chunk([b"aa", b"b"] * 50, None, "bp")
    bad = chunk.copy()
    bad[12 + 20:12 + 24] = np.frombuffer(np.uint32(1 << 30).tobytes(), dtype=np.uint8)
    with pytest.raises(B.BtrError) as e:
        api.string_dict_info(bad)
    assert e.value.status == 2
    with pytest.raises(B.BtrError):
        api.string_dict_info(chunk[:20])
    with pytest.raises(pyoracle.OracleError):
        pyoracle.string_dict_stream(bad, pyoracle.CODES)
