"""GPU: column part files streamed through the part pipeline (pinned read buffer -> copy-in ->
decode -> copy-out, consecutive parts overlapped), compared with the CPU restatement."""
import ctypes as C

import numpy as np
import pytest
import torch

import btrblocks_amd as B
from btrblocks_amd import _native as N
from btrblocks_amd import api, files, synth
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def _bits(a):
    return a.view(np.uint32 if a.dtype == np.int32 else np.uint64)


def _oracle_column(path, column, parts, want_nullmap=False):
    vals, nms = [], []
    for p in range(parts):
        image = np.fromfile(f"{path}/column{column}_part{p}", dtype=np.uint8)
        for o, n in B.part_chunks(image):
            v, m = pyoracle.decode_chunk(image[o:o + n], want_nullmap=True)
            vals.append(v)
            nms.append(m)
    return (np.concatenate(vals), np.concatenate(nms)) if want_nullmap else np.concatenate(vals)


def test_directory_of_many_parts_streams_through_the_pipeline(gpu_ctx, tmp_path, monkeypatch):
    monkeypatch.setattr(files, "PART_BYTES", 1 << 20)  # many small parts per column
    rows = 3_000_000 + 4321
    nm = synth.random_nullmap(rows, 0.9)
    cols = [synth.int_dict(rows), synth.double_decimal(rows), synth.int_rle(rows), synth.int_for(rows, seed=9)]
    schemes = ["auto", "pseudodecimal(bp,bp,uncompressed)", "rle(bp,bp)", "pfor"]
    _, part_counts = files.write_directory(str(tmp_path), cols, nullmaps=[nm, None, None, nm], scheme=schemes, threads=8)
    assert sorted(part_counts)[1] >= 2 and max(part_counts) >= 8
    d = files.BtrDirectory(str(tmp_path), ctx=gpu_ctx)
    for c, want in enumerate(cols):
        got, got_nm = d.decode_column(c, with_nullmap=True)
        ref, ref_nm = _oracle_column(str(tmp_path), c, part_counts[c], want_nullmap=True)
        np.testing.assert_array_equal(_bits(got), _bits(ref))
        np.testing.assert_array_equal(got_nm, ref_nm)
        valid = ref_nm.astype(bool)
        np.testing.assert_array_equal(_bits(got[valid]), _bits(want[valid]))
    st = d.pipeline.stats
    assert st.parts == sum(part_counts) and st.tuples == rows * len(cols)
    # a second pass over the same pipeline reuses its buffers
    again = d.decode_column(1)
    np.testing.assert_array_equal(_bits(again), _bits(cols[1]))


def test_pipeline_to_device_destinations_and_errors(gpu_ctx):
    rows = 4 * 65536 + 99
    values = synth.int_for(rows)
    image = B.compress_column(values, None, "for(bp)", threads=2)
    pipe = api.PartPipeline(gpu_ctx, slots=2)
    d_out = torch.zeros(rows * 4 + 64, dtype=torch.uint8, device="cuda:0")
    for _ in range(5):  # more submits than slots: slots are recycled
        buf = pipe.acquire(image.size)
        buf[:] = image
        pipe.submit(buf, dev_dst=d_out.data_ptr(), dst_bytes=rows * 4)
    assert pipe.drain() == 0
    np.testing.assert_array_equal(d_out[:rows * 4].cpu().numpy().view(np.int32), values)
    # destination too small, missing destination
    with pytest.raises(B.BtrError) as e:
        pipe.submit(image, dev_dst=d_out.data_ptr(), dst_bytes=1000)
    assert e.value.status == 4
    with pytest.raises(B.BtrError) as e:
        pipe.submit(image, dst_bytes=rows * 4)
    assert e.value.status == 1
    # a header the host walk rejects fails the submit; the pipeline stays usable
    bad = image.copy()
    o, n = B.part_chunks(image)[1]
    bad[o + 12 + 4] = 77  # FOR's next-scheme code
    with pytest.raises(B.BtrError) as e:
        pipe.submit(bad, dev_dst=d_out.data_ptr(), dst_bytes=rows * 4)
    assert e.value.status in (2, 3)
    # a payload only the kernels can find malformed is reported by drain, with the part's index
    bad = image.copy()
    words = bad[o + 12 + 5:o + 12 + 9].view(np.uint32)  # BP u32_count
    bad[o + 12 + 5 + 5 + 0:o + 12 + 5 + 5 + 4] = 0xFF   # first bit-width header word
    out = api.PinnedBuffer(rows * 4 + 16)
    pipe.submit(image, host_dst=out.ptr, dst_bytes=rows * 4)
    try:
        pipe.submit(bad, host_dst=out.ptr, dst_bytes=rows * 4)
        with pytest.raises(B.BtrError) as e:
            pipe.drain()
        assert e.value.status == 2 and "part" in str(e.value)
    except B.BtrError as e:  # or the host walk already caught it
        assert e.status in (2, 3)
        pipe.drain()
    pipe.submit(image, host_dst=out.ptr, dst_bytes=rows * 4)
    assert pipe.drain() == 0
    np.testing.assert_array_equal(out.array[:rows * 4].view(np.int32), values)
    del words
    pipe.close()
