/* btrblocks_amd - C ABI of the MI355X (gfx950) BtrBlocks decompression engine.
 *
 * The reference (maxi-k/btrblocks) has no FFI; its decode boundary is the C++ API
 *   IntegerScheme::decompress / DoubleScheme::decompress   btrblocks/scheme/CompressionScheme.hpp:37-41, 70-74
 *   BtrReader::readColumn + getters                        btrblocks/compression/BtrReader.hpp:12-45
 *   Datablock::decompress / Datablock::compress            btrblocks/compression/Datablock.hpp:57-72
 * Each entry point below names the reference interface it stands in for. All pointers are plain
 * host or device addresses; no C++ or torch types cross this boundary.
 *
 * Decoding runs on the GPU only. Every decode entry point fails with BTR_ERR_DEVICE when no HIP
 * device or no gfx950 code object is available; there is no CPU fallback. Compression and scheme
 * selection stay on the CPU (btr_compress_*).
 */
#ifndef BTRBLOCKS_AMD_H
#define BTRBLOCKS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum btr_status {
  BTR_OK = 0,
  BTR_ERR_INVALID_ARGUMENT = 1,
  BTR_ERR_CORRUPT = 2,      /* header/offset inconsistent with the supplied byte length          */
  BTR_ERR_UNSUPPORTED = 3,  /* scheme without a decoder (the reference's Truncation), string columns */
  BTR_ERR_BUFFER_TOO_SMALL = 4,
  BTR_ERR_DEVICE = 5,       /* HIP runtime error, no device, or no gfx950 code object            */
  BTR_ERR_MISALIGNED = 6,   /* bit-packed payload not 4-byte aligned at its device address       */
  BTR_ERR_OUT_OF_MEMORY = 7
} btr_status;

/* Message of the last error on the calling thread (never NULL). */
const char* btr_last_error(void);
const char* btr_version(void);

/* Column types and nullmap types: btrblocks/common/Units.hpp:45-57. */
enum { BTR_COLUMN_INTEGER = 0, BTR_COLUMN_DOUBLE = 1, BTR_COLUMN_STRING = 2 };
enum { BTR_NULLMAP_ALLONES = 0, BTR_NULLMAP_ALLZEROS = 1, BTR_NULLMAP_REGULAR = 2, BTR_NULLMAP_FLIPPED = 3 };

/* ------------------------------------------------------------------------------------------
 * Host-side inspection of compressed images (no GPU needed).
 * ------------------------------------------------------------------------------------------ */
typedef struct btr_chunk_info {
  uint32_t tuple_count;   /* BtrReader::getTupleCount,          BtrReader.hpp:35-37  */
  uint8_t column_type;    /* BtrReader::getColumnType,          BtrReader.hpp:38     */
  uint8_t scheme;         /* ColumnChunkMeta::compression_type, Datablock.hpp:13     */
  uint8_t nullmap_type;   /* ColumnChunkMeta::nullmap_type,     Datablock.hpp:14     */
  uint8_t reserved;
  uint64_t decoded_bytes; /* BtrReader::getDecompressedDataSize, BtrReader.cpp:186-207 */
} btr_chunk_info;

/* One ColumnChunkMeta chunk (compression/Datablock.hpp:12-21). */
btr_status btr_chunk_inspect(const void* chunk, size_t bytes, btr_chunk_info* out);
/* Scheme tree of a chunk in text form, e.g. "RLE(DICT(BP),BP)": BtrReader::getSchemeDescription,
 * BtrReader.cpp:76-97. Writes a NUL-terminated string of at most cap bytes. */
btr_status btr_chunk_describe(const void* chunk, size_t bytes, char* out, size_t cap);
/* Column part image {u32 num_chunks; u32 offsets[]} + chunks (storage/Chunk.hpp:95-98):
 * BtrReader::getChunkCount / getChunkMetadata, BtrReader.hpp:28-39. */
btr_status btr_part_chunk_count(const void* part, size_t bytes, uint32_t* out);
btr_status btr_part_chunk(const void* part, size_t bytes, uint32_t index, size_t* offset, size_t* length);
/* Envelopes of chunks [first, first + count) of a part image in one call: what a BtrReader answers
 * its getters from (BtrReader.hpp:28-39, BtrReader.cpp:159-207). Any of infos / offsets / lengths
 * may be NULL. */
btr_status btr_part_inspect(const void* part, size_t bytes, uint32_t first, uint32_t count,
                            btr_chunk_info* infos, uint64_t* offsets, uint64_t* lengths);
/* Relation-level block (DatablockMeta + ColumnMeta[], compression/Datablock.hpp:37-55). */
btr_status btr_datablock_inspect(const void* block, size_t bytes, uint32_t* tuple_count, uint32_t* column_count);
btr_status btr_datablock_column(const void* block, size_t bytes, uint32_t column, btr_chunk_info* out);

/* String column chunk in the DICT scheme: the fields of DynamicDictionaryStructure
 * (scheme/string/DynamicDictionary.hpp:7-18) with every offset rebased to the chunk start, plus the
 * run count and child schemes of the codes' RLE node when use_rle_optimized_path is set
 * (what RLE::decompressRuns reads, scheme/templated/RLE.hpp:111-128). Only the integer sub-streams
 * of such a chunk are decoded by this library; the dictionary strings (FSST or plain) are not. */
typedef struct btr_string_dict_info {
  uint32_t tuple_count;
  uint32_t total_decompressed_size;
  uint32_t num_codes;              /* dictionary entries */
  uint8_t use_fsst;
  uint8_t use_rle_optimized_path;
  uint8_t codes_scheme;            /* integer scheme code of the codes stream   */
  uint8_t lengths_scheme;          /* integer scheme code of the lengths stream */
  uint32_t runs_count;             /* RLE-optimised path only, else 0 */
  uint8_t run_values_scheme;
  uint8_t run_lengths_scheme;
  uint8_t reserved[2];
  uint32_t pool_offset;            /* DynamicDictionaryStructure::data */
  uint32_t codes_offset;
  uint32_t lengths_offset;         /* 0 without FSST */
  uint32_t fsst_offsets_offset;    /* 0 without FSST */
  uint32_t run_values_offset;      /* RLE-optimised path only, else 0 */
  uint32_t run_lengths_offset;
} btr_string_dict_info;
btr_status btr_string_dict_inspect(const void* chunk, size_t bytes, btr_string_dict_info* out);

/* ------------------------------------------------------------------------------------------
 * Compression (CPU). Stands in for Datablock::compress(const InputChunk&, u8*)
 * (compression/Datablock.cpp:66-148) and Datablock::compress(const Chunk&, BytesArray&)
 * (:192-338). `scheme_spec` is "auto" (NULL = auto) or a forced cascade such as
 * "rle(dict(bp),bp)", "for(bp)", "pseudodecimal(bp,bp,uncompressed)".
 * ------------------------------------------------------------------------------------------ */
typedef struct btr_compress_options {
  uint8_t max_cascade_depth; /* 0 = default 3 (btrblocks.hpp:31,37) */
  uint8_t enable_for;        /* the reference's ENABLE_FOR_SCHEME build flag */
  uint8_t reserved[6];
} btr_compress_options;

size_t btr_compress_bound(uint8_t column_type, uint32_t tuple_count);
btr_status btr_compress_chunk(uint8_t column_type, const void* values, const uint8_t* nullmap,
                              uint32_t tuple_count, const char* scheme_spec,
                              const btr_compress_options* options, void* out, size_t out_capacity,
                              size_t* out_bytes);
typedef struct btr_column_input {
  uint8_t column_type;
  const void* values;
  const uint8_t* nullmap; /* byte per row, 1 = valid; NULL = all valid */
  const char* scheme_spec;
} btr_column_input;
btr_status btr_compress_datablock(const btr_column_input* columns, uint32_t column_count,
                                  uint32_t tuple_count, const btr_compress_options* options,
                                  void* out, size_t out_capacity, size_t* out_bytes);

/* Splits a column into blocks of `block_rows` rows (0 = 65536, btrblocks.hpp:24; the split of
 * Relation::getRanges, storage/Relation.cpp:42-58), compresses each block as one chunk on
 * `threads` host threads (0 = all) and writes a column part image (storage/Chunk.cpp:250-295).
 * Unlike the reference's 16 MiB part files, the image holds all chunks of the column. */
btr_status btr_compress_column(uint8_t column_type, const void* values, const uint8_t* nullmap,
                               uint64_t rows, uint32_t block_rows, const char* scheme_spec,
                               const btr_compress_options* options, int threads, void* out,
                               size_t out_capacity, size_t* out_bytes);
size_t btr_compress_column_bound(uint8_t column_type, uint64_t rows, uint32_t block_rows);

/* One block of a string column as a DICT chunk (strings::DynamicDictionary::compress,
 * scheme/string/DynamicDictionary.cpp:42-208). `offsets` holds tuple_count + 1 byte offsets into
 * `bytes`. The codes are written as RLE with use_rle_optimized_path when the average run is
 * longer than 3 rows (:183-189), otherwise as `codes_spec` says. The fsst library is not part of
 * this build: with fsst_layout != 0 the pool is laid out as the FSST variant is (symbol-table area,
 * string bytes, u32 offsets, lengths stream) but the strings are stored verbatim and the
 * symbol-table area is zeroed, so such a chunk exercises the lengths sub-stream only. */
size_t btr_compress_string_dict_bound(uint64_t string_bytes, uint32_t tuple_count);
btr_status btr_compress_string_dict_chunk(const uint8_t* bytes, const uint32_t* offsets, uint32_t tuple_count,
                                          const uint8_t* nullmap, const char* codes_spec,
                                          const char* lengths_spec, int fsst_layout,
                                          const btr_compress_options* options, void* out,
                                          size_t out_capacity, size_t* out_bytes);

/* ------------------------------------------------------------------------------------------
 * GPU decode.
 * ------------------------------------------------------------------------------------------ */
typedef struct btr_gpu_ctx btr_gpu_ctx;
typedef struct btr_gpu_plan btr_gpu_plan;

/* Binds to a HIP device. Fails with BTR_ERR_DEVICE if there is none. */
btr_status btr_gpu_ctx_create(int device, btr_gpu_ctx** out);
void btr_gpu_ctx_destroy(btr_gpu_ctx* ctx);

/* One column chunk to decode. The compressed bytes must be resident on the device at `dev_src`
 * (16-byte aligned, as in column part files) and readable on the host at `host_src`: headers are
 * walked on the host once, when the plan is built. */
typedef struct btr_chunk_desc {
  const void* host_src;
  const void* dev_src;
  size_t src_bytes;
  void* dev_dst;            /* tuple_count * 4 (integer) or * 8 (double) bytes, 8-byte aligned */
  size_t dst_bytes;
  uint8_t* dev_nullmap_dst; /* tuple_count bytes (BitmapWrapper::writeBITMAP), or NULL */
} btr_chunk_desc;

/* Builds the device-resident decode plan for n chunks: the counterpart of constructing
 * BtrReader objects over column part images (BtrReader.cpp:12-15) - parsing, validation and
 * table placement happen here, not in btr_gpu_decode.
 * The chunks may belong to any number of columns: a plan over all columns of a relation (or of a
 * rank's block range of it) is how tools/conversion/decompression-speed.cpp:45-58 hands every
 * column x part x chunk to one scheduler. Such a plan launches one kernel per kernel kind and
 * occupancy class, not one per column, beside each other on the caller's stream and streams of
 * the context. */
btr_status btr_gpu_plan_create(btr_gpu_ctx* ctx, const btr_chunk_desc* chunks, size_t n, btr_gpu_plan** out);

/* Plan over chunks [first_chunk, first_chunk + chunk_count) of a column part image resident on the
 * device at dev_part: the counterpart of constructing one BtrReader over the part
 * (BtrReader.hpp:14, BtrReader.cpp:12-15) and calling readColumn for every chunk. Chunk k of the
 * range decodes to dev_dst + the 16-byte-aligned sizes of the chunks before it; its byte-per-row
 * nullmap (dev_nullmap_dst may be NULL) to dev_nullmap_dst + the rows before it. */
btr_status btr_gpu_plan_create_part(btr_gpu_ctx* ctx, const void* host_part, const void* dev_part, size_t bytes,
                                    uint32_t first_chunk, uint32_t chunk_count, void* dev_dst, size_t dst_bytes,
                                    uint8_t* dev_nullmap_dst, size_t nullmap_bytes, btr_gpu_plan** out);

/* Plan over one relation-level block: every integer/double column becomes one work item
 * (Datablock::decompress, Datablock.cpp:340-417). dsts[c] / nullmap_dsts[c] are device pointers
 * (nullmap_dsts or its entries may be NULL). */
btr_status btr_gpu_plan_create_datablock(btr_gpu_ctx* ctx, const void* host_block, const void* dev_block,
                                         size_t bytes, void* const* dsts, uint8_t* const* nullmap_dsts,
                                         uint32_t column_count, btr_gpu_plan** out);

/* Integer sub-streams of string DICT chunks, decoded with the integer kernels:
 *   CODES       the code of every row:      codes_scheme.decompress(...), DynamicDictionary.cpp:381-387
 *   LENGTHS     the length of every entry:  lengths_scheme.decompress(...), :444-451 (FSST chunks)
 *   RUN_VALUES / RUN_LENGTHS  the (code, run length) pairs of the RLE-optimised path:
 *               rle.decompressRuns(...), :239-251 and :484-496, templated/RLE.hpp:111-128
 * Each item decodes one sub-stream of one chunk to int32 values at dev_dst (8-byte aligned; the
 * counts are tuple_count, num_codes and runs_count of btr_string_dict_inspect). */
enum { BTR_SUBSTREAM_CODES = 0, BTR_SUBSTREAM_LENGTHS = 1, BTR_SUBSTREAM_RUN_VALUES = 2, BTR_SUBSTREAM_RUN_LENGTHS = 3 };
typedef struct btr_substream_desc {
  const void* host_src;  /* the whole chunk, as in btr_chunk_desc */
  const void* dev_src;
  size_t src_bytes;
  uint32_t which;        /* BTR_SUBSTREAM_* */
  uint32_t reserved;
  void* dev_dst;
  size_t dst_bytes;
} btr_substream_desc;
btr_status btr_gpu_plan_create_string_dict(btr_gpu_ctx* ctx, const btr_substream_desc* items, size_t n,
                                           btr_gpu_plan** out);
void btr_gpu_plan_destroy(btr_gpu_plan* plan);

typedef struct btr_plan_stats {
  uint64_t chunks;
  uint64_t tuples;
  uint64_t decoded_bytes;    /* bytes written to the value columns                              */
  uint64_t nullmap_bytes;    /* bytes written to byte-per-row nullmaps                          */
  uint64_t needed_bytes;     /* compressed bytes the decoder reads (headers + live payload)     */
  uint64_t stored_bytes;     /* compressed bytes as stored (sum of src_bytes)                   */
  uint64_t scratch_bytes;    /* HBM scratch for tables that did not fit in LDS                  */
  uint32_t max_lds_bytes;    /* dynamic LDS per workgroup                                       */
  uint32_t max_depth;
  uint32_t kernel_mask;      /* bit k set: chunks of the plan are decoded by kernel k (btr_kernel_name);
                                bits 16 and 17: stream chunks of several kinds share a mixed launch */
  uint32_t launches;         /* kernel launches per btr_gpu_decode                              */
  double compile_seconds;    /* host time of the header walk + program compilation              */
  double upload_seconds;     /* host time of laying out and uploading programs and work items   */
} btr_plan_stats;
btr_status btr_gpu_plan_stats(const btr_gpu_plan* plan, btr_plan_stats* out);
/* Name of the HIP kernel behind bit k of btr_plan_stats::kernel_mask (as profilers show it), or
 * NULL when k is out of range. */
const char* btr_kernel_name(uint32_t k);

/* Enqueues the decode of every chunk of the plan on `hip_stream` (a hipStream_t, NULL = default
 * stream). Asynchronous; allocates nothing. Replaces the scheme.decompress(...) call inside
 * BtrReader::readColumn (BtrReader.cpp:45-71) for all chunks at once. */
btr_status btr_gpu_decode(btr_gpu_ctx* ctx, btr_gpu_plan* plan, void* hip_stream);

/* Enqueues the decode of several plans (typically the columns of one relation) so that they run
 * beside each other: the plans are spread over a few streams owned by the context, which fork from
 * `hip_stream` and join it again by events; work queued behind the call sees every plan decoded.
 * Asynchronous; allocates nothing. Stands in for the tbb::parallel_for_each over columns in
 * tools/conversion/decompression-speed.cpp:45-58. Launches that are too small to fill the GPU
 * (a rank's shard of a relation, the last round of a large launch) overlap this way; the plans'
 * destinations must not overlap. Two launches of the same plan must not overlap either. */
btr_status btr_gpu_decode_columns(btr_gpu_ctx* ctx, btr_gpu_plan* const* plans, size_t n, void* hip_stream);

/* btr_gpu_plan_create + btr_gpu_decode (+ btr_gpu_plan_check) in one call that allocates nothing
 * on the device or in the HIP runtime: no hipMalloc / hipFree / hipHostMalloc, no stream or event
 * creation. (On the host it compiles the chunks' programs into ordinary heap memory, on the
 * calling thread only.) It has no counterpart in the reference; it
 * exists for callers that cannot allocate - tight serving loops and harnesses that forbid
 * hipMalloc on the decode path. Inputs and outputs are the caller's device buffers, as in
 * btr_gpu_plan_create. Everything a plan would allocate on the device (work items, programs,
 * status words, tables that do not fit in LDS) lives in one 16 MiB static array per GPU, so a
 * call takes at most 256 chunks whose programs fit in 992 KiB and whose spilled tables fit in
 * 15 MiB (BTR_ERR_INVALID_ARGUMENT / BTR_ERR_BUFFER_TOO_SMALL beyond that). Every call on a GPU
 * shares that GPU's array: calls that overlap on one GPU race (a call may start once the previous
 * one's work on the stream has completed). With status_or == NULL the call only enqueues; otherwise
 * it waits for the stream and returns BTR_ERR_CORRUPT when a kernel reported a malformed stream. */
btr_status btr_gpu_decode_chunks_noalloc(btr_gpu_ctx* ctx, const btr_chunk_desc* chunks, size_t n, void* hip_stream,
                                         uint32_t* status_or);

/* Waits for the stream and reports malformed streams found by the kernels: *status_or is the OR
 * of all per-chunk status words (0 = clean), *first_bad the first chunk with a non-zero word.
 * Returns BTR_ERR_CORRUPT when any is set - the counterpart of the reference's
 * Generic_Exception("Decompressing XPBP failed") (integer/PBP.cpp:44-47). */
btr_status btr_gpu_plan_check(btr_gpu_ctx* ctx, btr_gpu_plan* plan, void* hip_stream,
                              uint32_t* status_or, uint64_t* first_bad);

/* Test hooks (no counterpart in the reference). btr_gpu_plan_debug_access reads (write == 0) or
 * overwrites `bytes` bytes at `offset` of one of a plan's device tables - region 0: work items,
 * 1: launch entries, 2: program blob - after a device synchronisation; the tests use it to hand a
 * workgroup a launch entry or program that is not the one the host compiled and to check that the
 * decode ends with BTR_ERR_CORRUPT (status bit 0x80) instead of interpreting it.
 * btr_debug_launch_entry_check is the check word of a launch entry {item, prog_off, prog_bytes}. */
btr_status btr_gpu_plan_debug_access(btr_gpu_plan* plan, uint32_t region, size_t offset, void* host, size_t bytes,
                                     int write);
uint32_t btr_debug_launch_entry_check(uint32_t item, uint32_t prog_off, uint32_t prog_bytes);
/* Test hook that needs no device: compiles the chunks as btr_gpu_plan_create would (only host_src is
 * read; dev_src and dev_dst are taken as addresses, nothing is launched) and describes the launches
 * of the plan, one text line per launch in launch order:
 *   "<kernel name> threads=<t> lds=<bytes> chunks=<n> [<first block>:<kernel kind name> ...] items=<i,j,...>"
 * `text` receives at most text_bytes - 1 characters and a terminating 0; BTR_ERR_BUFFER_TOO_SMALL if
 * the description does not fit. */
btr_status btr_debug_launch_layout(const btr_chunk_desc* chunks, size_t n, char* text, size_t text_bytes);

/* Host-in / host-out decode of one chunk: BtrReader::readColumn (BtrReader.cpp:31-74) with the
 * nullmap of BitmapWrapper::writeBITMAP (extern/RoaringBitmap.cpp:32-53). Copies the chunk to the
 * device, decodes, copies the column back. `nullmap_dst` may be NULL. */
btr_status btr_gpu_decode_chunk_host(btr_gpu_ctx* ctx, const void* chunk, size_t bytes, void* dst,
                                     size_t dst_bytes, uint8_t* nullmap_dst);

/* ------------------------------------------------------------------------------------------
 * Streaming decode of column parts that start in host memory.
 * Stands in for the loop of tools/conversion/decompression-speed.cpp:160-170 + 205-238 (read every
 * column<c>_part<p> file, construct a BtrReader over it, readColumn every chunk) when the parts do
 * not fit on the device at once or arrive one by one. Per part: header walk and program
 * compilation on the host, part image host -> device, decode, decoded column device -> host; the
 * four stages of consecutive parts overlap on three HIP streams. A pipeline owns `slots` (0 = 3)
 * sets of device buffers, grown on demand and reused.
 * ------------------------------------------------------------------------------------------ */
typedef struct btr_gpu_pipeline btr_gpu_pipeline;
typedef struct btr_part_job {
  const void* host_part;     /* column part image (storage/Chunk.hpp:95-98); pinned memory copies fastest */
  size_t bytes;
  void* host_dst;            /* decoded chunks back to back, every chunk start 16-byte aligned; or NULL */
  void* dev_dst;             /* non-NULL: decode straight to this device buffer (host_dst is then optional) */
  size_t dst_bytes;
  uint8_t* host_nullmap_dst; /* byte per row, chunks back to back; or NULL */
  uint8_t* dev_nullmap_dst;  /* or NULL */
  size_t nullmap_bytes;
} btr_part_job;
typedef struct btr_pipeline_stats {
  uint64_t parts, chunks, tuples;
  uint64_t stored_bytes;     /* part bytes copied in      */
  uint64_t decoded_bytes;    /* column bytes decoded      */
  double compile_seconds;    /* host time in the header walk + program compilation           */
  double wait_seconds;       /* host time blocked on a slot whose previous part was in flight */
} btr_pipeline_stats;
btr_status btr_gpu_pipeline_create(btr_gpu_ctx* ctx, uint32_t slots, btr_gpu_pipeline** out);
void btr_gpu_pipeline_destroy(btr_gpu_pipeline* pipeline);
/* Pinned host buffer of the slot the next submit will use, at least `bytes` long: read the part
 * file straight into it, then pass it as btr_part_job::host_part. Blocks while that slot's
 * previous part is still in flight. Optional - any host pointer may be submitted, but it must stay
 * untouched until the part has been copied (btr_gpu_pipeline_drain, or `slots` later submits). */
btr_status btr_gpu_pipeline_acquire(btr_gpu_pipeline* pipeline, size_t bytes, void** host_buffer);
/* Enqueues one part; returns once its copies and launches are queued. Destinations must stay
 * valid until btr_gpu_pipeline_drain. */
btr_status btr_gpu_pipeline_submit(btr_gpu_pipeline* pipeline, const btr_part_job* job);
/* Waits for every submitted part. *status_or is the OR of the kernels' status words since the last
 * drain, *first_bad_part the submit index of the first part with a non-zero word;
 * BTR_ERR_CORRUPT when any is set (as btr_gpu_plan_check). */
btr_status btr_gpu_pipeline_drain(btr_gpu_pipeline* pipeline, uint32_t* status_or, uint64_t* first_bad_part);
btr_status btr_gpu_pipeline_stats(const btr_gpu_pipeline* pipeline, btr_pipeline_stats* out);
/* Pinned host memory (hipHostMalloc / hipHostFree) for part images and decoded columns. */
btr_status btr_host_alloc(size_t bytes, void** out);
btr_status btr_host_free(void* ptr);

/* Thin wrappers over hipMalloc / hipFree / hipMemcpy for callers without a HIP toolchain. */
btr_status btr_gpu_malloc(btr_gpu_ctx* ctx, size_t bytes, void** out);
btr_status btr_gpu_free(btr_gpu_ctx* ctx, void* ptr);
btr_status btr_gpu_upload(btr_gpu_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
btr_status btr_gpu_download(btr_gpu_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
btr_status btr_gpu_synchronize(btr_gpu_ctx* ctx, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* BTRBLOCKS_AMD_H */
