"""Host logic of a plan's launches (c_api.hip::lay_out) through the device-free test hook
btr_debug_launch_layout: which chunks share a launch, in what order, with how much LDS."""
import ctypes as C

import numpy as np

import btrblocks_amd as B
from btrblocks_amd import _native as N, synth

ROWS = 3 * 65536


def _layout(columns):
    """columns: [(generator name, scheme)] -> (launch lines, item index ranges per column)"""
    lib = N.load()
    images, descs, ranges = [], [], []
    at = 0
    for gen, spec in columns:
        image = B.compress_column(getattr(synth, gen)(ROWS), None, spec, threads=2)
        images.append(image)
        chunks = B.part_chunks(image)
        ranges.append(range(at, at + len(chunks)))
        at += len(chunks)
        for off, n in chunks:
            d = N.ChunkDesc()
            d.host_src = image.ctypes.data + off
            d.dev_src = image.ctypes.data + off   # an address with the alignment of the host bytes; never read
            d.src_bytes = n
            d.dev_dst = 1 << 20                   # never written: the hook launches nothing
            d.dst_bytes = 1 << 30
            d.dev_nullmap_dst = None
            descs.append(d)
    arr = (N.ChunkDesc * len(descs))(*descs)
    text = C.create_string_buffer(1 << 16)
    N.check(lib.btr_debug_launch_layout(arr, len(descs), text, len(text)))
    lines = []
    for line in text.value.decode().splitlines():
        head, items = line.rsplit(" items=", 1)
        name = head.split()[0]
        fields = dict(f.split("=") for f in head.split()[1:4])
        segs = head[head.index("[") + 1:head.index("]")].split()
        lines.append({"kernel": name, "threads": int(fields["threads"]), "lds": int(fields["lds"]), "chunks": int(fields["chunks"]),
                      "segments": [(int(s.split(":")[0]), s.split(":")[1]) for s in segs],
                      "items": [int(x) for x in items.split(",")]})
    return lines, ranges


def test_one_column_is_one_launch_in_block_order():
    # run tables of different sizes per block: some blocks of this column cross an occupancy step
    lines, ranges = _layout([("int_rle_pool", "rle(dict(bp),bp)")])
    assert len(lines) == 1 and lines[0]["kernel"] == "btr_decode_int_kernel"
    assert lines[0]["items"] == list(ranges[0])


def test_stream_kinds_share_one_mixed_launch_longest_blocks_first():
    lines, ranges = _layout([("int_for", "for(bp)"), ("double_decimal", "pseudodecimal(bp,bp,uncompressed)"),
                             ("int_dict", "dict(bp)")])
    assert len(lines) == 1
    lg = lines[0]
    assert lg["kernel"] == "btr_stream_mixed_kernel" and lg["threads"] == 512 and lg["chunks"] == 9
    assert [s[1] for s in lg["segments"]] == ["btr_stream_pseudo_kernel", "btr_stream_dict_lds_kernel", "btr_stream_add_kernel"]
    assert [s[0] for s in lg["segments"]] == [0, 3, 6]
    assert lg["items"] == list(ranges[1]) + list(ranges[2]) + list(ranges[0])
    assert lg["lds"] <= 40 * 1024 + 512  # four workgroups per CU whatever the kind


def test_relation_launches_by_kernel_and_occupancy_class():
    cols = [("int_rle", "rle(bp,bp)"), ("int_for", "for(bp)"), ("int_dict", "dict(bp)"), ("int_outliers", "pfor"),
            ("int_constant", "onevalue"), ("double_decimal", "pseudodecimal(bp,bp,uncompressed)"),
            ("double_decimal_runs", "pseudodecimal(rle(dict(bp),bp),rle(bp,bp),uncompressed)"),
            ("double_dict", "dict(bp)"), ("double_runs", "rle(uncompressed,bp)")]
    lines, ranges = _layout(cols)
    by_kernel = {}
    for lg in lines:
        by_kernel.setdefault(lg["kernel"], []).append(lg)
    # every chunk is in exactly one launch
    assert sorted(i for lg in lines for i in lg["items"]) == list(range(sum(len(r) for r in ranges)))
    assert len(by_kernel["btr_stream_mixed_kernel"]) == 1 and by_kernel["btr_stream_mixed_kernel"][0]["chunks"] == 9
    assert len(by_kernel["btr_stream_dictd_kernel"]) == 1 and by_kernel["btr_stream_dictd_kernel"][0]["threads"] == 1024
    assert len(by_kernel["btr_decode_int_pfor_kernel"]) == 1
    # the nested Pseudodecimal (three bitmaps: three workgroups per CU) and the RLE doubles (four) do not share a launch
    dbl = by_kernel["btr_decode_dbl_kernel"]
    assert len(dbl) == 2
    assert sorted(tuple(lg["items"]) for lg in dbl) == sorted([tuple(ranges[6]), tuple(ranges[8])])
    assert max(lg["lds"] for lg in dbl) > 40 * 1024 > min(lg["lds"] for lg in dbl)
    # RLE and the fill share the generic integer launch, column by column
    ints = by_kernel["btr_decode_int_kernel"]
    assert len(ints) == 1 and ints[0]["items"] == list(ranges[0]) + list(ranges[4])
    # launches come longest first (by estimated time): the last one is not the largest
    assert lines[0]["chunks"] >= lines[-1]["chunks"]


def test_hook_reports_a_short_buffer():
    lib = N.load()
    image = B.compress_column(synth.int_for(65536), None, "for(bp)")
    off, n = B.part_chunks(image)[0]
    d = N.ChunkDesc()
    d.host_src = d.dev_src = image.ctypes.data + off
    d.src_bytes, d.dev_dst, d.dst_bytes, d.dev_nullmap_dst = n, 1 << 20, 1 << 30, None
    arr = (N.ChunkDesc * 1)(d)
    text = C.create_string_buffer(8)
    assert lib.btr_debug_launch_layout(arr, 1, text, len(text)) == 4  # BTR_ERR_BUFFER_TOO_SMALL
