"""TEST INFRASTRUCTURE ONLY: ctypes binding of the CPU oracle (oracle/btr_oracle.cpp).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg - never by the
btrblocks_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "btr_oracle.cpp")
LIB = os.path.join(_HERE, "_build", "libbtr_oracle.so")
_lib = None


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", SRC, "-o", LIB])
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        lib.btr_oracle_error.restype = C.c_char_p
        lib.btr_oracle_decode_chunk.restype = C.c_int
        lib.btr_oracle_decode_chunk.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.btr_oracle_decode_datablock_column.restype = C.c_int
        lib.btr_oracle_decode_datablock_column.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                                           C.c_size_t, C.c_void_p]
        lib.btr_oracle_decompression_speed.restype = C.c_double
        lib.btr_oracle_decompression_speed.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t,
                                                       C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        lib.btr_oracle_hardware_threads.restype = C.c_int
        lib.btr_oracle_string_dict_stream.restype = C.c_int
        lib.btr_oracle_string_dict_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t,
                                                      C.POINTER(C.c_uint32)]
        lib.btr_oracle_string_dict_rows.restype = C.c_int
        lib.btr_oracle_string_dict_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                                    C.c_size_t, C.POINTER(C.c_uint64)]
        _lib = lib
    return _lib


class OracleError(RuntimeError):
    pass


def _u8(buf):
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    return np.ascontiguousarray(a.view(np.uint8))


def decode_chunk(chunk, want_nullmap=False):
    """Decodes one ColumnChunkMeta chunk on the CPU. Returns values (and the byte-per-row nullmap)."""
    lib = load()
    a = _u8(chunk)
    if a.size < 12:
        raise OracleError("chunk shorter than its envelope")
    n = int(a[8:12].view(np.uint32)[0])
    col = int(a[2])
    out = np.empty(n, dtype=np.float64 if col == 1 else np.int32)
    nm = np.empty(n, dtype=np.uint8) if want_nullmap else None
    rc = lib.btr_oracle_decode_chunk(a.ctypes.data, a.size, out.ctypes.data, out.nbytes,
                                     nm.ctypes.data if want_nullmap else None)
    if rc != 0:
        raise OracleError(lib.btr_oracle_error().decode())
    return (out, nm) if want_nullmap else out


def decode_datablock(block):
    """[(values, nullmap)] per column of a relation-level block."""
    lib = load()
    a = _u8(block)
    n = int(a[0:4].view(np.uint32)[0])
    cols = int(a[8:12].view(np.uint32)[0])
    res = []
    for c in range(cols):
        col_type = int(a[16 + 16 * c + 1])
        out = np.empty(n, dtype=np.float64 if col_type == 1 else np.int32)
        nm = np.empty(n, dtype=np.uint8)
        rc = lib.btr_oracle_decode_datablock_column(a.ctypes.data, a.size, c, out.ctypes.data, out.nbytes,
                                                    nm.ctypes.data)
        if rc != 0:
            raise OracleError(lib.btr_oracle_error().decode())
        res.append((out, nm))
    return res


CODES, LENGTHS, RUN_VALUES, RUN_LENGTHS = 0, 1, 2, 3


def string_dict_stream(chunk, which):
    """One integer sub-stream (CODES / LENGTHS / RUN_VALUES / RUN_LENGTHS) of a string DICT chunk."""
    lib = load()
    a = _u8(chunk)
    count = C.c_uint32()
    n = int(a[8:12].view(np.uint32)[0]) if a.size >= 12 else 0
    out = np.empty(max(n, 1), dtype=np.int32)
    rc = lib.btr_oracle_string_dict_stream(a.ctypes.data, a.size, which, out.ctypes.data, out.size, C.byref(count))
    if rc != 0:
        raise OracleError(lib.btr_oracle_error().decode())
    return out[:count.value].copy()


def string_dict_rows(chunk):
    """(offsets[n + 1], bytes) of the rows of a string DICT chunk without FSST."""
    lib = load()
    a = _u8(chunk)
    total = C.c_uint64()
    rc = lib.btr_oracle_string_dict_rows(a.ctypes.data, a.size, None, 0, None, 0, C.byref(total))
    if rc != 0:
        raise OracleError(lib.btr_oracle_error().decode())
    n = int(a[8:12].view(np.uint32)[0])
    offsets = np.empty(n + 1, dtype=np.uint32)
    text = np.empty(max(total.value, 1), dtype=np.uint8)
    rc = lib.btr_oracle_string_dict_rows(a.ctypes.data, a.size, offsets.ctypes.data, offsets.size, text.ctypes.data,
                                         text.size, C.byref(total))
    if rc != 0:
        raise OracleError(lib.btr_oracle_error().decode())
    return offsets, text[:total.value].copy()


def decompression_speed(image, chunks, threads=0, reps=1):
    """Restated decompression-speed over the chunks [(offset, length)] of a part image.
    Returns (seconds per pass, decompressed bytes per pass, threads used)."""
    lib = load()
    a = _u8(image)
    n = len(chunks)
    ptrs = (C.c_void_p * n)(*[a.ctypes.data + o for o, _ in chunks])
    sizes = (C.c_size_t * n)(*[ln for _, ln in chunks])
    out_bytes = C.c_uint64()
    if threads <= 0:
        threads = lib.btr_oracle_hardware_threads()
    sec = lib.btr_oracle_decompression_speed(ptrs, sizes, n, threads, reps, C.byref(out_bytes))
    if sec < 0:
        raise OracleError("oracle decode failed during the speed run")
    return sec, out_bytes.value, threads
