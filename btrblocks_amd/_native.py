"""ctypes binding of libbtrblocks_amd.so (the C ABI declared in include/btrblocks_amd.h).

The library is the product: HIP decode kernels for gfx950 plus the CPU compressor. There is no
Python or CPU decode path here; every decode call goes through the shared library and fails
loudly when it (or a GPU) is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BTR_LIB_PATH: load another build of the same library (the diagnostic build of tools/gpu_dbg.sh)
LIB_PATH = os.environ.get("BTR_LIB_PATH") or os.path.join(_HERE, "libbtrblocks_amd.so")

BTR_OK = 0
STATUS_NAMES = {
    0: "BTR_OK",
    1: "BTR_ERR_INVALID_ARGUMENT",
    2: "BTR_ERR_CORRUPT",
    3: "BTR_ERR_UNSUPPORTED",
    4: "BTR_ERR_BUFFER_TOO_SMALL",
    5: "BTR_ERR_DEVICE",
    6: "BTR_ERR_MISALIGNED",
    7: "BTR_ERR_OUT_OF_MEMORY",
}

COLUMN_INTEGER, COLUMN_DOUBLE, COLUMN_STRING = 0, 1, 2


class BtrError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class ChunkInfo(C.Structure):
    _fields_ = [
        ("tuple_count", C.c_uint32),
        ("column_type", C.c_uint8),
        ("scheme", C.c_uint8),
        ("nullmap_type", C.c_uint8),
        ("reserved", C.c_uint8),
        ("decoded_bytes", C.c_uint64),
    ]


class CompressOptions(C.Structure):
    _fields_ = [
        ("max_cascade_depth", C.c_uint8),
        ("enable_for", C.c_uint8),
        ("reserved", C.c_uint8 * 6),
    ]


class ColumnInput(C.Structure):
    _fields_ = [
        ("column_type", C.c_uint8),
        ("values", C.c_void_p),
        ("nullmap", C.c_void_p),
        ("scheme_spec", C.c_char_p),
    ]


class ChunkDesc(C.Structure):
    _fields_ = [
        ("host_src", C.c_void_p),
        ("dev_src", C.c_void_p),
        ("src_bytes", C.c_size_t),
        ("dev_dst", C.c_void_p),
        ("dst_bytes", C.c_size_t),
        ("dev_nullmap_dst", C.c_void_p),
    ]


class StringDictInfo(C.Structure):
    _fields_ = [
        ("tuple_count", C.c_uint32),
        ("total_decompressed_size", C.c_uint32),
        ("num_codes", C.c_uint32),
        ("use_fsst", C.c_uint8),
        ("use_rle_optimized_path", C.c_uint8),
        ("codes_scheme", C.c_uint8),
        ("lengths_scheme", C.c_uint8),
        ("runs_count", C.c_uint32),
        ("run_values_scheme", C.c_uint8),
        ("run_lengths_scheme", C.c_uint8),
        ("reserved", C.c_uint8 * 2),
        ("pool_offset", C.c_uint32),
        ("codes_offset", C.c_uint32),
        ("lengths_offset", C.c_uint32),
        ("fsst_offsets_offset", C.c_uint32),
        ("run_values_offset", C.c_uint32),
        ("run_lengths_offset", C.c_uint32),
    ]


SUBSTREAM_CODES, SUBSTREAM_LENGTHS, SUBSTREAM_RUN_VALUES, SUBSTREAM_RUN_LENGTHS = 0, 1, 2, 3


class SubstreamDesc(C.Structure):
    _fields_ = [
        ("host_src", C.c_void_p),
        ("dev_src", C.c_void_p),
        ("src_bytes", C.c_size_t),
        ("which", C.c_uint32),
        ("reserved", C.c_uint32),
        ("dev_dst", C.c_void_p),
        ("dst_bytes", C.c_size_t),
    ]


class PlanStats(C.Structure):
    _fields_ = [
        ("chunks", C.c_uint64),
        ("tuples", C.c_uint64),
        ("decoded_bytes", C.c_uint64),
        ("nullmap_bytes", C.c_uint64),
        ("needed_bytes", C.c_uint64),
        ("stored_bytes", C.c_uint64),
        ("scratch_bytes", C.c_uint64),
        ("max_lds_bytes", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("kernel_mask", C.c_uint32),
        ("launches", C.c_uint32),
        ("compile_seconds", C.c_double),
        ("upload_seconds", C.c_double),
    ]


class PartJob(C.Structure):
    _fields_ = [
        ("host_part", C.c_void_p),
        ("bytes", C.c_size_t),
        ("host_dst", C.c_void_p),
        ("dev_dst", C.c_void_p),
        ("dst_bytes", C.c_size_t),
        ("host_nullmap_dst", C.c_void_p),
        ("dev_nullmap_dst", C.c_void_p),
        ("nullmap_bytes", C.c_size_t),
    ]


class PipelineStats(C.Structure):
    _fields_ = [
        ("parts", C.c_uint64),
        ("chunks", C.c_uint64),
        ("tuples", C.c_uint64),
        ("stored_bytes", C.c_uint64),
        ("decoded_bytes", C.c_uint64),
        ("compile_seconds", C.c_double),
        ("wait_seconds", C.c_double),
    ]


# Every symbol include/btrblocks_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "btr_last_error": (C.c_char_p, []),
    "btr_version": (C.c_char_p, []),
    "btr_chunk_inspect": (C.c_int, [_P, C.c_size_t, C.POINTER(ChunkInfo)]),
    "btr_chunk_describe": (C.c_int, [_P, C.c_size_t, C.c_char_p, C.c_size_t]),
    "btr_part_chunk_count": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_uint32)]),
    "btr_part_chunk": (C.c_int, [_P, C.c_size_t, C.c_uint32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "btr_part_inspect": (C.c_int, [_P, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(ChunkInfo), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64)]),
    "btr_datablock_inspect": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "btr_datablock_column": (C.c_int, [_P, C.c_size_t, C.c_uint32, C.POINTER(ChunkInfo)]),
    "btr_string_dict_inspect": (C.c_int, [_P, C.c_size_t, C.POINTER(StringDictInfo)]),
    "btr_compress_string_dict_bound": (C.c_size_t, [C.c_uint64, C.c_uint32]),
    "btr_compress_string_dict_chunk": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_char_p, C.c_char_p, C.c_int,
                                                 C.POINTER(CompressOptions), _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "btr_gpu_plan_create_string_dict": (C.c_int, [_P, C.POINTER(SubstreamDesc), C.c_size_t, C.POINTER(_P)]),
    "btr_compress_bound": (C.c_size_t, [C.c_uint8, C.c_uint32]),
    "btr_compress_chunk": (C.c_int, [C.c_uint8, _P, _P, C.c_uint32, C.c_char_p, C.POINTER(CompressOptions), _P,
                                     C.c_size_t, C.POINTER(C.c_size_t)]),
    "btr_compress_datablock": (C.c_int, [C.POINTER(ColumnInput), C.c_uint32, C.c_uint32,
                                         C.POINTER(CompressOptions), _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "btr_compress_column": (C.c_int, [C.c_uint8, _P, _P, C.c_uint64, C.c_uint32, C.c_char_p,
                                      C.POINTER(CompressOptions), C.c_int, _P, C.c_size_t,
                                      C.POINTER(C.c_size_t)]),
    "btr_compress_column_bound": (C.c_size_t, [C.c_uint8, C.c_uint64, C.c_uint32]),
    "btr_gpu_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "btr_gpu_ctx_destroy": (None, [_P]),
    "btr_gpu_plan_create": (C.c_int, [_P, C.POINTER(ChunkDesc), C.c_size_t, C.POINTER(_P)]),
    "btr_gpu_plan_create_part": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_uint32, C.c_uint32, _P, C.c_size_t, _P,
                                           C.c_size_t, C.POINTER(_P)]),
    "btr_gpu_plan_create_datablock": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(_P), C.POINTER(_P),
                                                C.c_uint32, C.POINTER(_P)]),
    "btr_gpu_plan_destroy": (None, [_P]),
    "btr_gpu_plan_stats": (C.c_int, [_P, C.POINTER(PlanStats)]),
    "btr_kernel_name": (C.c_char_p, [C.c_uint32]),
    "btr_gpu_decode": (C.c_int, [_P, _P, _P]),
    "btr_gpu_decode_columns": (C.c_int, [_P, C.POINTER(C.c_void_p), C.c_size_t, _P]),
    "btr_gpu_decode_chunks_noalloc": (C.c_int, [_P, C.POINTER(ChunkDesc), C.c_size_t, _P, C.POINTER(C.c_uint32)]),
    "btr_gpu_plan_check": (C.c_int, [_P, _P, _P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "btr_gpu_decode_chunk_host": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "btr_gpu_plan_debug_access": (C.c_int, [_P, C.c_uint32, C.c_size_t, _P, C.c_size_t, C.c_int]),
    "btr_debug_launch_entry_check": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "btr_debug_launch_layout": (C.c_int, [_P, C.c_size_t, C.c_char_p, C.c_size_t]),
    "btr_gpu_pipeline_create": (C.c_int, [_P, C.c_uint32, C.POINTER(_P)]),
    "btr_gpu_pipeline_destroy": (None, [_P]),
    "btr_gpu_pipeline_acquire": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "btr_gpu_pipeline_submit": (C.c_int, [_P, C.POINTER(PartJob)]),
    "btr_gpu_pipeline_drain": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "btr_gpu_pipeline_stats": (C.c_int, [_P, C.POINTER(PipelineStats)]),
    "btr_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(_P)]),
    "btr_host_free": (C.c_int, [_P]),
    "btr_gpu_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "btr_gpu_free": (C.c_int, [_P, _P]),
    "btr_gpu_upload": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "btr_gpu_download": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "btr_gpu_synchronize": (C.c_int, [_P, _P]),
}

_lib = None


def load():
    """Loads the shared library once. torch (when importable) is imported first so that both share
    one HIP runtime and device pointers/streams are interchangeable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m btrblocks_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for decoding.")
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != BTR_OK:
        raise BtrError(status, load().btr_last_error().decode("utf-8", "replace"))
