"""Directories of compressed columns in the reference's on-disk layout.

The reference's tools write one directory per relation (tools/conversion/btrtocsv.cpp,
decompression-speed.cpp:113-170):
  metadata                 FileMetadata{u32 num_columns; u32 num_chunks} followed by one
                           ColumnPartInfo{u8 type; 3 pad bytes; u32 num_parts} per column
                           (compression/Datablock.hpp:23-35, Datablock.cpp:27-52)
  column<c>_part<p>        a column part: {u32 num_chunks; u32 offsets[]} then 16-byte aligned
                           ColumnChunkMeta chunks (storage/Chunk.cpp:250-295); a part takes chunks
                           while it stays within 16 MiB (storage/Chunk.cpp:12, 228-243)
A part file is exactly the "part image" BtrReader takes, so reading is np.fromfile + BtrReader.
"""
import os
import struct

import numpy as np

from . import api

PART_BYTES = 16 * 1024 * 1024  # chunk_part_size_threshold
_ALIGN = 16


def _pack_part(chunks):
    """{u32 num_chunks; u32 offsets[]} + chunks, each starting on a 16-byte boundary of the file."""
    head = 4 + 4 * len(chunks)
    out = bytearray(head)
    struct.pack_into("<I", out, 0, len(chunks))
    for i, c in enumerate(chunks):
        out.extend(b"\0" * (-len(out) % _ALIGN))
        struct.pack_into("<I", out, 4 + 4 * i, len(out))
        out.extend(c.tobytes())
    return bytes(out)


def write_directory(path, columns, nullmaps=None, scheme="auto", block_rows=65536, threads=0):
    """Compresses `columns` (numpy int32 / float64 arrays of equal length) block by block and writes
    the relation directory. Returns (num_chunks, parts per column)."""
    os.makedirs(path, exist_ok=True)
    nullmaps = nullmaps or [None] * len(columns)
    schemes = scheme if isinstance(scheme, (list, tuple)) else [scheme] * len(columns)
    rows = len(columns[0])
    part_counts, types = [], []
    for c, (values, nm, spec) in enumerate(zip(columns, nullmaps, schemes)):
        assert len(values) == rows, "columns of a relation have the same length"
        image = api.compress_column(values, nm, spec, block_rows=block_rows, threads=threads)
        chunk_ranges = api.part_chunks(image)
        parts, cur, cur_bytes = [], [], 0
        for off, n in chunk_ranges:
            if n > PART_BYTES:
                raise ValueError("a compressed chunk larger than the 16 MiB part limit cannot be stored")
            if cur and cur_bytes + n > PART_BYTES:
                parts.append(cur)
                cur, cur_bytes = [], 0
            cur.append(image[off:off + n])
            cur_bytes += n
        if cur or not parts:
            parts.append(cur)
        for p, chunks in enumerate(parts):
            with open(os.path.join(path, f"column{c}_part{p}"), "wb") as f:
                f.write(_pack_part(chunks))
        part_counts.append(len(parts))
        types.append(api._column_type(values))
    num_chunks = (rows + block_rows - 1) // block_rows
    with open(os.path.join(path, "metadata"), "wb") as f:
        f.write(struct.pack("<II", len(columns), num_chunks))
        for t, k in zip(types, part_counts):
            f.write(struct.pack("<B3xI", t, k))
    return num_chunks, part_counts


def read_metadata(path):
    """-> (num_chunks, [(column_type, num_parts), ...])"""
    raw = open(os.path.join(path, "metadata"), "rb").read()
    if len(raw) < 8:
        raise ValueError("metadata file too short")
    num_columns, num_chunks = struct.unpack_from("<II", raw, 0)
    if len(raw) < 8 + 8 * num_columns:
        raise ValueError("metadata file too short for its column count")
    cols = [struct.unpack_from("<B3xI", raw, 8 + 8 * c) for c in range(num_columns)]
    return num_chunks, cols


class BtrDirectory:
    """A relation directory opened for GPU decoding: one BtrReader per column part."""

    def __init__(self, path, ctx=None, device=0):
        from .reader import BtrReader
        self.path = path
        self.ctx = ctx or api.GpuContext(device)
        self.num_chunks, self.columns = read_metadata(path)
        self.readers = []
        for c, (_, parts) in enumerate(self.columns):
            self.readers.append([
                BtrReader(np.fromfile(os.path.join(path, f"column{c}_part{p}"), dtype=np.uint8), ctx=self.ctx)
                for p in range(parts)])

    def decode_column(self, column):
        """Decodes every part of a column on the GPU; returns the host array."""
        parts = []
        for r in self.readers[column]:
            r.decodeAll()
            parts.append(r.column_host())
        if not parts:
            return np.empty(0, dtype=np.int32)
        return np.concatenate(parts)
