"""btrblocks_amd: MI355X (gfx950) decompression engine for BtrBlocks column chunks.

Layout:
  csrc/         HIP decode kernels, host plan compiler, CPU compressor, the C ABI
  _native.py    ctypes binding of libbtrblocks_amd.so
  api.py        numpy/torch level wrappers (compress_*, GpuContext, DecodePlan)
  reader.py     mirror of the reference's BtrReader / Datablock decode interface
  files.py      relation directories in the reference's on-disk layout (metadata + column<c>_part<p>)
  synth.py      deterministic synthetic columns (the reference's test/benchmark distributions)
"""
from ._native import BtrError, COLUMN_DOUBLE, COLUMN_INTEGER  # noqa: F401
from .api import (DecodePlan, GpuContext, chunk_info, compress_chunk, compress_column,  # noqa: F401
                  compress_datablock, datablock_columns, describe, part_chunks)
from .reader import BtrReader, Datablock  # noqa: F401
from .files import BtrDirectory, read_metadata, write_directory  # noqa: F401
