"""numpy / torch level wrappers over the C ABI.

Host-side helpers (inspection, CPU compression) work on numpy arrays. GPU decode works on device
buffers; torch is used only to own device memory and to name the current HIP stream.
"""
import ctypes as C

import numpy as np

from . import _native as N

_DTYPE = {N.COLUMN_INTEGER: np.int32, N.COLUMN_DOUBLE: np.float64}


def _as_u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    if a.dtype != np.uint8:
        a = a.view(np.uint8)
    return np.ascontiguousarray(a)


def _column_type(values):
    if values.dtype == np.int32:
        return N.COLUMN_INTEGER
    if values.dtype == np.float64:
        return N.COLUMN_DOUBLE
    raise TypeError("columns must be int32 or float64 numpy arrays")


def _options(max_cascade_depth=0, enable_for=False):
    o = N.CompressOptions()
    o.max_cascade_depth = max_cascade_depth
    o.enable_for = 1 if enable_for else 0
    return o


def _nullmap_ptr(nullmap, n):
    if nullmap is None:
        return None, None
    nm = np.ascontiguousarray(nullmap, dtype=np.uint8)
    if nm.shape != (n,):
        raise ValueError("nullmap must hold one byte per row")
    return nm, nm.ctypes.data


# ---------------------------------------------------------------------------------------------
# inspection
# ---------------------------------------------------------------------------------------------
def chunk_info(chunk):
    lib = N.load()
    a = _as_u8(chunk)
    info = N.ChunkInfo()
    N.check(lib.btr_chunk_inspect(a.ctypes.data, a.size, C.byref(info)))
    return info


def describe(chunk):
    """Scheme tree of a chunk as text, e.g. 'RLE(DICT(BP),BP)' (BtrReader::getSchemeDescription)."""
    lib = N.load()
    a = _as_u8(chunk)
    out = C.create_string_buffer(1024)
    N.check(lib.btr_chunk_describe(a.ctypes.data, a.size, out, len(out)))
    return out.value.decode()


def part_inspect(part):
    """Envelopes of every chunk of a column part image in one call: (infos, offsets, lengths) with
    infos a ctypes array of ChunkInfo and offsets / lengths uint64 numpy arrays."""
    lib = N.load()
    a = _as_u8(part)
    n = C.c_uint32()
    N.check(lib.btr_part_chunk_count(a.ctypes.data, a.size, C.byref(n)))
    infos = (N.ChunkInfo * n.value)()
    offsets = np.zeros(n.value, dtype=np.uint64)
    lengths = np.zeros(n.value, dtype=np.uint64)
    N.check(lib.btr_part_inspect(a.ctypes.data, a.size, 0, n.value, infos,
                                 offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                 lengths.ctypes.data_as(C.POINTER(C.c_uint64))))
    return infos, offsets, lengths


def part_chunks(part):
    """[(offset, length)] of the chunks of a column part image."""
    _, offsets, lengths = part_inspect(part)
    return list(zip(offsets.tolist(), lengths.tolist()))


def string_dict_info(chunk):
    """Header fields of a string column chunk in the DICT scheme, offsets from the chunk start."""
    lib = N.load()
    a = _as_u8(chunk)
    info = N.StringDictInfo()
    N.check(lib.btr_string_dict_inspect(a.ctypes.data, a.size, C.byref(info)))
    return info


def datablock_columns(block):
    lib = N.load()
    a = _as_u8(block)
    n, cols = C.c_uint32(), C.c_uint32()
    N.check(lib.btr_datablock_inspect(a.ctypes.data, a.size, C.byref(n), C.byref(cols)))
    infos = []
    for c in range(cols.value):
        info = N.ChunkInfo()
        N.check(lib.btr_datablock_column(a.ctypes.data, a.size, c, C.byref(info)))
        infos.append(info)
    return n.value, infos


# ---------------------------------------------------------------------------------------------
# compression (CPU)
# ---------------------------------------------------------------------------------------------
def compress_chunk(values, nullmap=None, scheme="auto", max_cascade_depth=0, enable_for=False):
    """One ColumnChunkMeta chunk as a uint8 array (Datablock::compress(const InputChunk&))."""
    lib = N.load()
    values = np.ascontiguousarray(values)
    ctype = _column_type(values)
    n = values.size
    nm, nm_ptr = _nullmap_ptr(nullmap, n)
    cap = lib.btr_compress_bound(ctype, n)
    out = np.empty(cap, dtype=np.uint8)
    got = C.c_size_t()
    opt = _options(max_cascade_depth, enable_for)
    N.check(lib.btr_compress_chunk(ctype, values.ctypes.data, nm_ptr, n, scheme.encode(), C.byref(opt),
                                   out.ctypes.data, cap, C.byref(got)))
    return out[:got.value].copy()


def compress_column(values, nullmap=None, scheme="auto", block_rows=65536, threads=0,
                    max_cascade_depth=0, enable_for=False):
    """Column part image holding every 64k-row block of the column as one chunk."""
    lib = N.load()
    values = np.ascontiguousarray(values)
    ctype = _column_type(values)
    n = values.size
    nm, nm_ptr = _nullmap_ptr(nullmap, n)
    cap = lib.btr_compress_column_bound(ctype, n, block_rows)
    out = np.empty(cap, dtype=np.uint8)
    got = C.c_size_t()
    opt = _options(max_cascade_depth, enable_for)
    N.check(lib.btr_compress_column(ctype, values.ctypes.data, nm_ptr, n, block_rows, scheme.encode(),
                                    C.byref(opt), threads, out.ctypes.data, cap, C.byref(got)))
    return out[:got.value].copy()


def compress_string_dict_chunk(strings, nullmap=None, codes_scheme="auto", lengths_scheme="auto", fsst_layout=False):
    """One block of a string column as a DICT chunk (strings::DynamicDictionary::compress).
    `strings` is a sequence of bytes objects, or an (offsets[n + 1] uint32, bytes uint8) pair."""
    lib = N.load()
    if isinstance(strings, tuple):
        offsets, text = strings
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        text = np.ascontiguousarray(text, dtype=np.uint8)
    else:
        lens = np.fromiter((len(s) for s in strings), dtype=np.uint32, count=len(strings))
        offsets = np.zeros(len(strings) + 1, dtype=np.uint32)
        np.cumsum(lens, out=offsets[1:])
        text = np.frombuffer(b"".join(strings), dtype=np.uint8)
    n = offsets.size - 1
    text = text if text.size else np.zeros(1, dtype=np.uint8)
    nm, nm_ptr = _nullmap_ptr(nullmap, n)
    cap = lib.btr_compress_string_dict_bound(int(offsets[-1]), n)
    out = np.empty(cap, dtype=np.uint8)
    got = C.c_size_t()
    N.check(lib.btr_compress_string_dict_chunk(text.ctypes.data, offsets.ctypes.data, n, nm_ptr, codes_scheme.encode(),
                                               lengths_scheme.encode(), 1 if fsst_layout else 0, None,
                                               out.ctypes.data, cap, C.byref(got)))
    return out[:got.value].copy()


def compress_datablock(columns, max_cascade_depth=0, enable_for=False):
    """Relation-level block (Datablock::compress(const Chunk&, BytesArray&)).
    `columns`: list of (values, nullmap or None, scheme spec)."""
    lib = N.load()
    keep = []
    arr = (N.ColumnInput * len(columns))()
    n = None
    cap = 64
    for i, (values, nullmap, scheme) in enumerate(columns):
        values = np.ascontiguousarray(values)
        if n is None:
            n = values.size
        if values.size != n:
            raise ValueError("all columns of a block must have the same row count")
        nm, nm_ptr = _nullmap_ptr(nullmap, n)
        spec = (scheme or "auto").encode()
        keep.append((values, nm, spec))
        arr[i].column_type = _column_type(values)
        arr[i].values = values.ctypes.data
        arr[i].nullmap = nm_ptr
        arr[i].scheme_spec = spec
        cap += lib.btr_compress_bound(arr[i].column_type, n) + 32
    out = np.empty(cap, dtype=np.uint8)
    got = C.c_size_t()
    opt = _options(max_cascade_depth, enable_for)
    N.check(lib.btr_compress_datablock(arr, len(columns), n or 0, C.byref(opt), out.ctypes.data, cap,
                                       C.byref(got)))
    return out[:got.value].copy()


# ---------------------------------------------------------------------------------------------
# GPU decode
# ---------------------------------------------------------------------------------------------
class GpuContext:
    """btr_gpu_ctx: binds the library to one HIP device. Raises BtrError(BTR_ERR_DEVICE) when no
    GPU (or no gfx950 code object) is available."""

    def __init__(self, device=0):
        self.lib = N.load()
        self.device = device
        h = C.c_void_p()
        N.check(self.lib.btr_gpu_ctx_create(device, C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.btr_gpu_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def decode_chunk_host(self, chunk, want_nullmap=False):
        """Host in / host out: BtrReader::readColumn for one chunk."""
        a = _as_u8(chunk)
        info = chunk_info(a)
        out = np.empty(info.tuple_count, dtype=_DTYPE[info.column_type])
        nm = np.empty(info.tuple_count, dtype=np.uint8) if want_nullmap else None
        N.check(self.lib.btr_gpu_decode_chunk_host(self.handle, a.ctypes.data, a.size, out.ctypes.data,
                                                   out.nbytes, nm.ctypes.data if want_nullmap else None))
        return (out, nm) if want_nullmap else out


def decode_columns(ctx, plans, stream=None):
    """Decodes several plans (the columns of a relation) beside each other: btr_gpu_decode_columns,
    the counterpart of the parallel_for_each over columns in tools/conversion/decompression-speed.cpp:45-58."""
    handles = (C.c_void_p * len(plans))(*[p.handle for p in plans])
    N.check(ctx.lib.btr_gpu_decode_columns(ctx.handle, handles, len(plans), stream))


class DecodePlan:
    """btr_gpu_plan over chunks whose compressed bytes are already resident on the device."""

    def __init__(self, ctx, descs, keepalive=()):
        self.ctx = ctx
        self.lib = ctx.lib
        self._keep = keepalive
        arr = (N.ChunkDesc * len(descs))(*descs)
        h = C.c_void_p()
        N.check(self.lib.btr_gpu_plan_create(ctx.handle, arr, len(descs), C.byref(h)))
        self.handle = h

    @classmethod
    def for_part(cls, ctx, host_part, dev_ptr, first, count, dst_ptr, dst_bytes, nullmap_ptr=None, nullmap_bytes=0,
                 keepalive=()):
        """Plan over chunks [first, first + count) of a part image resident at dev_ptr
        (btr_gpu_plan_create_part: one call, outputs back to back)."""
        self = cls.__new__(cls)
        self.ctx, self.lib, self._keep = ctx, ctx.lib, keepalive
        a = _as_u8(host_part)
        h = C.c_void_p()
        N.check(self.lib.btr_gpu_plan_create_part(ctx.handle, a.ctypes.data, dev_ptr, a.size, first, count, dst_ptr,
                                                  dst_bytes, nullmap_ptr, nullmap_bytes, C.byref(h)))
        self.handle = h
        return self

    @classmethod
    def for_string_dict(cls, ctx, descs, keepalive=()):
        """Plan over integer sub-streams of string DICT chunks (btr_gpu_plan_create_string_dict);
        `descs` is a list of _native.SubstreamDesc."""
        self = cls.__new__(cls)
        self.ctx, self.lib, self._keep = ctx, ctx.lib, keepalive
        arr = (N.SubstreamDesc * len(descs))(*descs)
        h = C.c_void_p()
        N.check(self.lib.btr_gpu_plan_create_string_dict(ctx.handle, arr, len(descs), C.byref(h)))
        self.handle = h
        return self

    @classmethod
    def for_datablock(cls, ctx, host_block, dev_ptr, dsts, nullmap_dsts=None, keepalive=()):
        self = cls.__new__(cls)
        self.ctx, self.lib, self._keep = ctx, ctx.lib, keepalive
        a = _as_u8(host_block)
        n = len(dsts)
        d = (C.c_void_p * n)(*dsts)
        nm = (C.c_void_p * n)(*nullmap_dsts) if nullmap_dsts is not None else None
        h = C.c_void_p()
        N.check(self.lib.btr_gpu_plan_create_datablock(ctx.handle, a.ctypes.data, dev_ptr, a.size, d, nm, n,
                                                       C.byref(h)))
        self.handle = h
        return self

    @property
    def stats(self):
        s = N.PlanStats()
        N.check(self.lib.btr_gpu_plan_stats(self.handle, C.byref(s)))
        return s

    def decode(self, stream=None):
        N.check(self.lib.btr_gpu_decode(self.ctx.handle, self.handle, stream))

    def check(self, stream=None):
        st, first = C.c_uint32(), C.c_uint64()
        N.check(self.lib.btr_gpu_plan_check(self.ctx.handle, self.handle, stream, C.byref(st), C.byref(first)))
        return st.value

    def close(self):
        if getattr(self, "handle", None):
            self.lib.btr_gpu_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


class _PinnedMemory:
    """One btr_host_alloc allocation; freed when the last reference goes away."""

    def __init__(self, lib, nbytes):
        self.lib = lib
        p = C.c_void_p()
        N.check(lib.btr_host_alloc(max(int(nbytes), 1), C.byref(p)))
        self.ptr = p.value

    def __del__(self, _void_p=C.c_void_p):  # bound early: module globals may be gone at interpreter exit
        if getattr(self, "ptr", None):
            self.lib.btr_host_free(_void_p(self.ptr))
            self.ptr = None


class PinnedBuffer:
    """Pinned host memory (btr_host_alloc) seen as a uint8 numpy array. The memory belongs to
    `array` and to every view taken from it: it is freed when the PinnedBuffer and all of those
    arrays are gone, so a column returned as a view stays valid for as long as the caller holds it."""

    def __init__(self, nbytes):
        self.lib = N.load()
        self._mem = _PinnedMemory(self.lib, nbytes)
        self.ptr = self._mem.ptr
        self.nbytes = int(nbytes)
        raw = (C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr)
        raw._owner = self._mem  # numpy keeps `raw` as the base of the array and of its views
        self.array = np.ctypeslib.as_array(raw)[:self.nbytes]

    def close(self):
        """Drops this object's references; the memory itself goes with the last view of `array`."""
        self.array = None
        self._mem = None
        self.ptr = None

    def __del__(self):
        self.close()


class PartPipeline:
    """btr_gpu_pipeline: column parts in host memory -> decoded columns in host (or device) memory,
    with compile / copy-in / decode / copy-out of consecutive parts overlapped."""

    def __init__(self, ctx, slots=3):
        self.ctx = ctx
        self.lib = ctx.lib
        h = C.c_void_p()
        N.check(self.lib.btr_gpu_pipeline_create(ctx.handle, slots, C.byref(h)))
        self.handle = h

    def acquire(self, nbytes):
        """uint8 view of the next slot's pinned input buffer (at least nbytes long)."""
        p = C.c_void_p()
        N.check(self.lib.btr_gpu_pipeline_acquire(self.handle, nbytes, C.byref(p)))
        return np.ctypeslib.as_array((C.c_uint8 * max(nbytes, 1)).from_address(p.value))[:nbytes]

    def submit(self, part, host_dst=None, dev_dst=None, dst_bytes=0, host_nullmap=None, dev_nullmap=None,
               nullmap_bytes=0):
        """`part`: uint8 array holding the part image; destinations are raw addresses."""
        job = N.PartJob()
        job.host_part = part.ctypes.data
        job.bytes = part.size
        job.host_dst, job.dev_dst, job.dst_bytes = host_dst, dev_dst, dst_bytes
        job.host_nullmap_dst, job.dev_nullmap_dst, job.nullmap_bytes = host_nullmap, dev_nullmap, nullmap_bytes
        N.check(self.lib.btr_gpu_pipeline_submit(self.handle, C.byref(job)))

    def drain(self):
        st, first = C.c_uint32(), C.c_uint64()
        N.check(self.lib.btr_gpu_pipeline_drain(self.handle, C.byref(st), C.byref(first)))
        return st.value

    @property
    def stats(self):
        s = N.PipelineStats()
        N.check(self.lib.btr_gpu_pipeline_stats(self.handle, C.byref(s)))
        return s

    def close(self):
        if getattr(self, "handle", None):
            self.lib.btr_gpu_pipeline_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


def current_stream_ptr():
    """The raw hipStream_t of torch's current stream (0 = default stream)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
