"""Directories of compressed columns in the reference's on-disk layout.

The reference's tools write one directory per relation (tools/conversion/btrtocsv.cpp,
decompression-speed.cpp:113-170):
  metadata                 FileMetadata{u32 num_columns; u32 num_chunks} followed by one
                           ColumnPartInfo{u8 type; 3 pad bytes; u32 num_parts} per column
                           (compression/Datablock.hpp:23-35, Datablock.cpp:27-52)
  column<c>_part<p>        a column part: {u32 num_chunks; u32 offsets[]} then 16-byte aligned
                           ColumnChunkMeta chunks (storage/Chunk.cpp:250-295); a part takes chunks
                           while it stays within 16 MiB (storage/Chunk.cpp:12, 228-243)
A part file is exactly the "part image" BtrReader takes. BtrDirectory streams the part files through
the library's part pipeline: file p + 1 is read into a pinned buffer while part p is copied to the
GPU, decoded and copied back.
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
    """A relation directory opened for GPU decoding (the file loop of
    tools/conversion/decompression-speed.cpp:113-170)."""

    def __init__(self, path, ctx=None, device=0, slots=3):
        self.path = path
        self.ctx = ctx or api.GpuContext(device)
        self.num_chunks, self.columns = read_metadata(path)
        self.pipeline = api.PartPipeline(self.ctx, slots)
        self._readers = {}

    def part_path(self, column, part):
        return os.path.join(self.path, f"column{column}_part{part}")

    def _part_sizes(self, column):
        """(file bytes, decoded bytes with 16-byte aligned chunk starts, rows) per part, from the
        chunk envelopes alone: 12 bytes per chunk are read, not the payloads."""
        sizes = []
        for p in range(self.columns[column][1]):
            with open(self.part_path(column, p), "rb") as f:
                total = os.fstat(f.fileno()).st_size
                (n,) = struct.unpack("<I", f.read(4))
                if 4 + 4 * n > total:
                    raise ValueError("part file shorter than its chunk table")
                offsets = struct.unpack(f"<{n}I", f.read(4 * n))
                decoded = rows = 0
                pieces = []  # (offset inside the part's output, bytes) per chunk
                for off in offsets:
                    if off + 12 > total:
                        raise ValueError("chunk offset past the end of the part file")
                    f.seek(off)
                    env = f.read(12)
                    tuples = struct.unpack_from("<I", env, 8)[0]
                    decoded = (decoded + 15) // 16 * 16
                    pieces.append((decoded, tuples * (8 if env[2] == 1 else 4)))
                    decoded += pieces[-1][1]
                    rows += tuples
            sizes.append((total, decoded, rows, pieces))
        return sizes

    def decode_column(self, column, with_nullmap=False, out=None):
        """Streams every part of a column through the GPU. Returns the column as a host array: a
        view of pinned memory (a new allocation per call, or `out`, an api.PinnedBuffer) that keeps
        that memory alive for as long as the array or any view of it exists; with_nullmap adds the
        byte-per-row validity array."""
        ctype = self.columns[column][0]
        dtype = np.float64 if ctype == 1 else np.int32
        sizes = self._part_sizes(column)
        starts, at = [], 0
        for _, decoded, _, _ in sizes:
            at = (at + 15) // 16 * 16
            starts.append(at)
            at += decoded
        total_rows = sum(r for _, _, r, _ in sizes)
        out = out or api.PinnedBuffer(at + 16)
        if out.nbytes < at:
            raise ValueError("output buffer too small for the column")
        nm = api.PinnedBuffer(total_rows + 16) if with_nullmap else None
        row_at = 0
        for p, (nbytes, decoded, rows, _) in enumerate(sizes):
            buf = self.pipeline.acquire(nbytes)
            with open(self.part_path(column, p), "rb") as f:
                if f.readinto(memoryview(buf)) != nbytes:
                    raise IOError("short read of a column part file")
            self.pipeline.submit(buf, host_dst=out.ptr + starts[p], dst_bytes=decoded,
                                 host_nullmap=(nm.ptr + row_at) if nm else None, nullmap_bytes=rows if nm else 0)
            row_at += rows
        self.pipeline.drain()
        # Chunk starts are 16-byte aligned; full 64k-row blocks leave no gaps, so the usual column is
        # one contiguous view. Only a ragged block in the middle of a column needs the pieces joined.
        pieces = [(s + o, n) for s, (_, _, _, ps) in zip(starts, sizes) for o, n in ps]
        if all(a + n == b for (a, n), (b, _) in zip(pieces, pieces[1:])):
            values = out.array[:at].view(dtype)
        else:
            values = np.concatenate([out.array[a:a + n].view(dtype) for a, n in pieces])
        return (values, nm.array[:total_rows]) if with_nullmap else values

    def reader(self, column, part):
        """A device-resident BtrReader over one part (getters, readColumn of single chunks)."""
        from .reader import BtrReader
        key = (column, part)
        if key not in self._readers:
            self._readers[key] = BtrReader(np.fromfile(self.part_path(column, part), dtype=np.uint8), ctx=self.ctx)
        return self._readers[key]
