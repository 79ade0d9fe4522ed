"""Host-side mirror of the reference's reader interface for the decode path.

`BtrReader` keeps the method names and argument meaning of btrblocks::BtrReader
(btrblocks/compression/BtrReader.hpp:12-45): it is constructed over a column part image
({u32 num_chunks; u32 offsets[]} + chunks), answers the same getters from the chunk envelopes, and
`readColumn(index)` returns the decoded chunk. The image is uploaded to the GPU once, in the
constructor; decoding always runs through the HIP kernels behind the C ABI.
"""
import numpy as np

from . import _native as N
from . import api

_DTYPE = {N.COLUMN_INTEGER: np.int32, N.COLUMN_DOUBLE: np.float64}
_TYPE_NAME = {N.COLUMN_INTEGER: "integer", N.COLUMN_DOUBLE: "double", N.COLUMN_STRING: "string"}


class BtrReader:
    def __init__(self, data, ctx=None, device=0):
        import torch
        self._torch = torch
        self.ctx = ctx or api.GpuContext(device)
        self.device = torch.device("cuda", self.ctx.device)
        self.image = api._as_u8(data)
        self.infos, offsets, lengths = api.part_inspect(self.image)  # one call for all envelopes
        self.chunks = list(zip(offsets.tolist(), lengths.tolist()))
        self.d_image = torch.from_numpy(self.image).to(self.device)
        # decoded column: chunks back to back, each start 16-byte aligned (the layout
        # btr_gpu_plan_create_part uses)
        recs = np.frombuffer(self.infos, dtype=np.dtype([("tuple_count", "<u4"), ("t", "u1"), ("s", "u1"), ("n", "u1"),
                                                          ("r", "u1"), ("decoded_bytes", "<u8")]))
        padded = (recs["decoded_bytes"].astype(np.int64) + 15) // 16 * 16
        self.out_offsets = np.concatenate([[0], np.cumsum(padded)]).astype(np.int64)
        self.d_out = torch.empty(max(int(self.out_offsets[-1]), 16), dtype=torch.uint8, device=self.device)
        self.row_offsets = np.concatenate([[0], np.cumsum(recs["tuple_count"].astype(np.int64))]).astype(np.int64)
        self.d_nullmap = None
        self._plans = {}

    # ---- getters (BtrReader.hpp:20-39) -----------------------------------------------------
    def getChunkCount(self):
        return len(self.chunks)

    def getTupleCount(self, index):
        return self.infos[index].tuple_count

    def getColumnType(self):
        return _TYPE_NAME[self.infos[0].column_type]

    def getDecompressedSize(self, index):
        return self.infos[index].decoded_bytes

    def getDecompressedDataSize(self, index):
        return self.infos[index].decoded_bytes

    def getSchemeDescription(self, index):
        o, n = self.chunks[index]
        return api.describe(self.image[o:o + n])

    def getBasicSchemeDescription(self, index):
        return self.getSchemeDescription(index).split("(")[0]

    # ---- decode ----------------------------------------------------------------------------
    def chunk_desc(self, index, with_nullmap=False):
        """btr_chunk_desc of chunk `index` (compressed bytes in this reader's device image, output in its
        device column): for plans over the chunks of several readers."""
        o, n = self.chunks[index]
        d = N.ChunkDesc()
        d.host_src = self.image.ctypes.data + o
        d.dev_src = self.d_image.data_ptr() + o
        d.src_bytes = n
        d.dev_dst = self.d_out.data_ptr() + int(self.out_offsets[index])
        d.dst_bytes = self.infos[index].decoded_bytes
        d.dev_nullmap_dst = (self.d_nullmap.data_ptr() + int(self.row_offsets[index])) if with_nullmap else None
        return d

    _desc = chunk_desc

    def plan(self, indices=None, with_nullmap=False):
        """Decode plan over the given chunks (all by default); cached."""
        key = (tuple(indices) if indices is not None else None, with_nullmap)
        if key not in self._plans:
            if with_nullmap and self.d_nullmap is None:
                self.d_nullmap = self._torch.empty(max(int(self.row_offsets[-1]), 16), dtype=self._torch.uint8,
                                                   device=self.device)
            if indices is None:  # the whole part: one call, no per-chunk descriptors
                self._plans[key] = api.DecodePlan.for_part(
                    self.ctx, self.image, self.d_image.data_ptr(), 0, len(self.chunks), self.d_out.data_ptr(),
                    self.d_out.numel(), self.d_nullmap.data_ptr() if with_nullmap else None,
                    self.d_nullmap.numel() if with_nullmap else 0, keepalive=(self.d_image, self.d_out, self.d_nullmap))
            else:
                self._plans[key] = api.DecodePlan(self.ctx, [self.chunk_desc(i, with_nullmap) for i in indices])
        return self._plans[key]

    def readColumn(self, index, with_bitmap=False):
        """Decodes chunk `index` on the GPU and returns it as a host array
        (BtrReader::readColumn, BtrReader.cpp:31-74; requires_copy is always false for numbers)."""
        p = self.plan([index], with_bitmap)
        p.decode(api.current_stream_ptr())
        p.check(api.current_stream_ptr())
        info = self.infos[index]
        o = int(self.out_offsets[index])
        host = self.d_out[o:o + info.decoded_bytes].cpu().numpy().view(_DTYPE[info.column_type]).copy()
        if not with_bitmap:
            return host
        r0 = int(self.row_offsets[index])
        return host, self.d_nullmap[r0:r0 + info.tuple_count].cpu().numpy().copy()

    def getBitmap(self, index):
        """Byte-per-row validity of chunk `index` (BitmapWrapper::writeBITMAP)."""
        return self.readColumn(index, with_bitmap=True)[1]

    def decodeAll(self, with_nullmap=False, check=True):
        """Decodes every chunk in one launch; returns the device tensor holding the column."""
        p = self.plan(None, with_nullmap)
        s = api.current_stream_ptr()
        p.decode(s)
        if check:
            p.check(s)
        return self.d_out

    def column_host(self):
        """The whole decoded column as one host array (after decodeAll)."""
        raw = self.d_out.cpu().numpy()
        parts = []
        for i, info in enumerate(self.infos):
            o = int(self.out_offsets[i])
            parts.append(raw[o:o + info.decoded_bytes].view(_DTYPE[info.column_type]))
        return np.concatenate(parts) if parts else np.empty(0, dtype=np.int32)


class Datablock:
    """Decode side of btrblocks::Datablock (compression/Datablock.hpp:57-72) for relation-level
    blocks: decompress(block) returns one (values, bitmap) pair per column, like the Chunk the
    reference returns (storage/Chunk.hpp:12-70)."""

    def __init__(self, ctx=None, device=0):
        self.ctx = ctx or api.GpuContext(device)

    @staticmethod
    def compress(columns, **kw):
        return api.compress_datablock(columns, **kw)

    def decompress(self, block):
        import torch
        dev = torch.device("cuda", self.ctx.device)
        a = api._as_u8(block)
        n, infos = api.datablock_columns(a)
        d_block = torch.from_numpy(a).to(dev)
        outs = [torch.empty(max(i.decoded_bytes, 16), dtype=torch.uint8, device=dev) for i in infos]
        nms = [torch.empty(max(n, 16), dtype=torch.uint8, device=dev) for _ in infos]
        plan = api.DecodePlan.for_datablock(self.ctx, a, d_block.data_ptr(), [o.data_ptr() for o in outs],
                                            [m.data_ptr() for m in nms], keepalive=(d_block, outs, nms))
        s = api.current_stream_ptr()
        plan.decode(s)
        plan.check(s)
        result = []
        for info, o, m in zip(infos, outs, nms):
            vals = o[:info.decoded_bytes].cpu().numpy().view(_DTYPE[info.column_type]).copy()
            result.append((vals, m[:n].cpu().numpy().copy()))
        plan.close()
        return result
