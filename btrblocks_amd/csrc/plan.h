// Decode plan: the flat, recursion-free program the host compiles from a chunk's header tree and
// the HIP kernel interprets, one workgroup per chunk.
//
// The reference decodes a cascade by virtual recursion with per-level scratch vectors
// (btrblocks/scheme/CompressionScheme.hpp:37-41, btrblocks/common/Units.hpp:118-123). On the GPU
// the same tree is flattened into
//   * sequences (Seq): arrays that live in LDS, in a per-chunk HBM scratch slice, in the
//     compressed bytes themselves (raw dictionaries, UNCOMPRESSED children) or in the output;
//   * pipes (Pipe): random-access value generators "value at row i" = a base (bit-unpack, run
//     lookup, exception lookup, ...) followed by post-ops (dictionary gather, FOR bias);
//   * steps (Step): the preparation order - build bit-unpack indexes, expand Roaring bitmaps,
//     materialize short child streams (run tables, exception lists, patches) - ending with the
//     step that materializes the root pipe into the output column.
// Full-length child streams (dictionary codes, FOR deltas, Pseudodecimal digits/exponents) are
// never materialized: they are evaluated on the fly inside the root step.
#pragma once
#include <stdint.h>

namespace btrplan {

enum : uint8_t { LOC_LDS = 0, LOC_SCRATCH = 1, LOC_SRC = 2, LOC_DST = 3, LOC_NULLMAP = 4 };

struct Seq {
  uint32_t off;    // byte offset inside its location (LDS offsets are relative to the table base)
  uint32_t count;  // elements
  uint8_t loc;
  uint8_t esize;     // bytes per element: 1, 2, 4 or 8
  uint8_t must_lds;  // host-side allocation hint only
  uint8_t pad;
};

enum : uint8_t {
  B_SEQ = 0,     // a[0] = seq id                                   (UNCOMPRESSED, materialized)
  B_ONE = 1,     // a[0],a[1] = value bits                          (ONE_VALUE)
  B_BP = 2,      // FastBinaryPacking<32>+VariableByte stream       (BP)
  B_RLE = 3,     // run lookup through a start-flag bitmap          (RLE)
  B_FREQ = 4,    // top value + exception bitmap                    (FREQUENCY)
  B_DICTD = 5,   // double dictionary over an integer code pipe     (DICT, doubles)
  B_PSEUDO = 6,  // digits * 10^-e with patches                     (PSEUDODECIMAL)
  B_DBP = 7,     // FastBinaryPacking over 2n words = raw f64 bits  (DOUBLE_BP)
};
// Parameter slots per base
//   B_BP / B_DBP : a0 payload byte offset in the chunk (4-byte aligned device address),
//                  a1 stored word count, a2 seq id of the group index {word offset, header},
//                  a3 values packed by FastBinaryPacking (multiple of 32), a4 seq id of the
//                  decoded VariableByte tail (or NONE), a5 number of 128-value groups
//   B_RLE        : a0 seq id of the start-flag bitmap, a1 seq id of its rank table,
//                  a2 seq id of run values, a3 run count
//   B_FREQ       : a0,a1 top value bits, a2 seq id bitmap, a3 seq id rank, a4 seq id exceptions
//                  (or NONE), a5 cardinality
//   B_DICTD      : a0 pipe id of the codes, a1 seq id of the dictionary
//   B_PSEUDO     : a0 pipe id digits (or NONE), a1 pipe id exponents, a2 seq id patches (or
//                  NONE), a3 seq id exception bitmap, a4 seq id rank, a5 converted_count
constexpr uint32_t NONE = 0xFFFFFFFFu;

enum : uint8_t { POST_DICT = 1, POST_ADD = 2 };
struct PostOp {
  uint8_t type;
  uint8_t pad[3];
  uint32_t arg;  // POST_DICT: seq id of the dictionary; POST_ADD: bias (wrapping add)
};
constexpr int MAX_POST = 2;

struct Pipe {
  uint8_t base;
  uint8_t is_double;
  uint8_t n_post;
  uint8_t pad;
  uint32_t count;  // rows this pipe can produce
  uint32_t a[6];
  PostOp post[MAX_POST];
};

enum : uint8_t {
  S_BP_INDEX = 0,       // a0 pipe id: walk the group headers of a B_BP/B_DBP stream, decode its tail;
                        // a1 seq id of kWalkScratchBytes of LDS for the walker's jump tables (or NONE)
  S_MATERIALIZE = 1,    // a0 pipe id, a1 dst seq id
  S_BITMAP_CLEAR = 2,   // a1 seq bitmap
  S_BITMAP_SET = 3,     // a0 src byte offset, a1 seq bitmap, a2 element count, a3 first row of the
                        // container (key << 16), a4 bit count, a5 element kind (SET_*)
  S_BITMAP_STARTS = 4,  // a0 seq run lengths (s32), a1 seq bitmap, a2 seq rank (or NONE), a3 bit count, a4 runs:
                        // exclusive-scans the lengths, sets one bit per run start and, when no run reaches
                        // more than a few 64-row words, fills the rank table as well
  S_BITMAP_EQ = 5,      // a0 pipe id, a1 seq bitmap, a3 bit count, a4 value: bit i = (pipe[i] == value)
  S_BITMAP_RANK = 6,    // a0 1 = the bitmap holds run starts and S_BITMAP_STARTS may have filled the table already
                        // (else NONE), a1 seq bitmap, a2 seq rank, a3 bit count, a4 expected popcount (or NONE),
                        // a5 status bit to raise on a mismatch
  S_NULLMAP = 7,        // a0 nullmap type, a1 seq bitmap (REGULAR/FLIPPED, else NONE), a3 row count
  S_PFOR_DECODE = 9,    // see below
  S_STREAM_ROOT = 8,    // a0 pipe id (B_BP or B_PSEUDO base), a1 dst seq id, a2 seq id of the powers-of-ten
                        // table (B_PSEUDO), a3 seq id of the walker's jump tables: root step that stages the payload(s) through LDS rings,
                        // indexing and unpacking them batch by batch
};
// S_PFOR_DECODE = 9: decodes a SIMDFastPFor<8>+VariableByte stream into a sequence.
//   a0 payload byte offset in the chunk (4-byte aligned device address), a1 stored word count,
//   a2 word index (in the program's constant words) of the page descriptors followed by one word
//   "first word of the VariableByte tail", a3 page count | (1u << 31 for PFOR_DELTA: prefix sum
//   afterwards), a4 dst seq id, a5 seq id of kPforWorkBytes of LDS
// Page descriptor (kPforPageWords words): first value, value count, first word of the packed
//   blocks, byte offset and size of the byte container, then for every exception width k = 2..32
//   the first word of its packed values and, after all of those, their counts.
// Element kinds of S_BITMAP_SET: the four member encodings of CRoaring's native serialization.
enum : uint32_t { SET_ARRAY32 = 0, SET_ARRAY16 = 1, SET_BITSET = 2, SET_RUNS = 3 };
struct Step {
  uint8_t kind;
  uint8_t no_barrier;  // the next step depends on nothing this one writes: the workgroup runs on into it
  uint8_t pad[2];
  uint32_t a[6];
};

struct ProgramHeader {
  uint32_t n_steps;
  uint32_t n_bp_steps;  // the leading S_BP_INDEX steps, run one per wave
  uint32_t n_pipes;
  uint32_t n_seqs;
  uint32_t table_base;  // absolute LDS byte offset where table space starts (after the program copy
                        // and 64 bytes of reduction scratch)
  uint32_t consts_off;  // byte offset of the constant words from the start of the program
  uint32_t magic;       // kProgramMagic
  uint32_t prog_bytes;  // size of the whole program: must equal what the launch entry and the work item say
};
// A program is laid out as: ProgramHeader | Step[n_steps] | Pipe[n_pipes] | Seq[n_seqs] | u32 consts[]
// A workgroup checks the header it has copied before it interprets anything (loop bounds, table
// offsets and barrier counts all come out of the program): a launch entry or program that is not
// the one the host wrote ends the workgroup with ST_PROGRAM instead of a long or endless loop.
constexpr uint32_t kProgramMagic = 0x50525442u;      // "BTRP"
constexpr uint32_t kMaxProgramBytes = 64u * 1024u;   // programs are a few hundred bytes to a few KiB

struct WorkItem {
  uint64_t src;          // device address of the chunk envelope / block column payload base
  uint64_t dst;          // device address of the decoded column (n * 4 or n * 8 bytes)
  uint64_t nullmap_dst;  // device address of the byte-per-row nullmap, or 0
  uint64_t scratch;      // device address of this item's scratch slice, or 0
  uint32_t prog_off;     // byte offset of the program in the plan's program blob
  uint32_t prog_bytes;
  uint32_t tuple_count;
  uint32_t src_bytes;
  uint32_t debug_flags;  // diagnostics only (BTR_DEBUG_FLAGS): 8 = record step timestamps; in builds with
                         // -DBTR_DEBUG_STEPS also 1 = skip index walks, 2 = skip the root step,
                         // 4 = root step only, k << 8 = stop before step k; Pseudodecimal stream kernel:
                         // 16 = loader, walker and barriers only, 32 = no stores; with 8 the stream kernels
                         // also record how long each wave role was busy
  uint32_t flags;        // ITEM_* bits
  uint64_t debug_buf;    // flag 8: kDebugSlots 100 MHz timestamps of this item
};
constexpr uint32_t kDebugSlots = 32;
// One entry per workgroup of a launch, in launch order: which work item it decodes and where that
// item's program lies. The program's place is repeated here so that a workgroup can start copying
// its program as soon as it has its entry, beside the load of the work item (two dependent memory
// round trips at the start of a block instead of three).
struct LaunchEntry {
  uint32_t item;
  uint32_t prog_off;
  uint32_t prog_bytes;
  uint32_t check;  // launch_entry_check(item, prog_off, prog_bytes)
};
static_assert(sizeof(LaunchEntry) == 16, "one 16-byte scalar load");
// Mixes the three fields of a launch entry: an entry read from the wrong place (a stale or
// mis-strided launch order) fails the comparison before its item index or program place is used.
constexpr uint32_t launch_entry_check(uint32_t item, uint32_t prog_off, uint32_t prog_bytes) {
  return (item * 0x9E3779B1u) ^ (prog_off * 0x85EBCA6Bu) ^ (prog_bytes * 0xC2B2AE35u) ^ kProgramMagic;
}
// The item belongs to a small launch that runs on a side stream beside the plan's main launch: its
// waves raise their priority, or they would crawl among the main launch's waves and end long after it.
enum : uint32_t { ITEM_SIDE_LAUNCH = 1u << 0 };

// Device status bits, one word per work item.
enum : uint32_t {
  ST_BP_HEADER = 1u << 0,   // a bit width above 32 or a group running past the stream
  ST_BP_TAIL = 1u << 1,     // VariableByte tail does not hold the expected number of values
  ST_RLE_COUNTS = 1u << 2,  // a run length <= 0 or lengths not adding up to the row count
  ST_DICT_CODE = 1u << 3,   // dictionary code outside the dictionary
  ST_PSEUDO = 1u << 4,      // exception count differs from tuple_count - converted_count
  ST_ROARING = 1u << 5,     // member outside the row range or cardinality mismatch
  ST_EXPONENT = 1u << 6,    // Pseudodecimal exponent index above 22
  ST_PROGRAM = 1u << 7,     // launch entry, work item or decode program inconsistent (raised on the item's word;
                            // on word 0 of the plan when the launch entry itself cannot be trusted)
  ST_INTERNAL = 1u << 8,    // a bounded wait inside a kernel ran out: the output of that chunk is not valid
};

// Which kernel decodes a work item. The generic kernels interpret any program; the stream kernels
// take programs of the form [dictionary copy into LDS]* + S_STREAM_ROOT.
enum : uint32_t {
  KERNEL_GENERIC_INT = 0,
  KERNEL_GENERIC_DBL = 1,
  KERNEL_STREAM_NONE = 2,      // BP
  KERNEL_STREAM_ADD = 3,       // FOR(BP)
  KERNEL_STREAM_DICT_LDS = 4,  // DICT(BP), dictionary staged in LDS
  KERNEL_STREAM_DICT_GLB = 5,  // DICT(BP), dictionary read from the compressed bytes
  KERNEL_STREAM_PSEUDO = 6,    // PSEUDODECIMAL(BP,BP,raw patches) doubles: digits through the ring
  KERNEL_STREAM_DICTD = 7,     // DICT(BP) doubles, dictionary staged in LDS or read from the compressed bytes
  KERNEL_PFOR_INT = 8,         // the generic kernels again, with the S_PFOR_DECODE step compiled in (the
  KERNEL_PFOR_DBL = 9,         // plain ones leave it out: it costs every other program registers)
  // 10..15: the stream kernels 2..7 with the nullmap steps behind the root compiled in (kept apart
  // for the same reason: registers)
  KERNEL_STREAM_NULLMAP = 8,   // added to a stream kind
  KERNEL_KINDS = 16,
};
constexpr bool kernel_is_generic(uint32_t k) {
  return k <= KERNEL_GENERIC_DBL || k == KERNEL_PFOR_INT || k == KERNEL_PFOR_DBL;
}

// A launch may decode chunks of several stream kinds: its workgroups are grouped into segments of
// one kind each, and a workgroup finds its kind from its index (the mixed stream kernels).
constexpr uint32_t kMaxSegments = 6;
struct LaunchSegments {
  uint32_t n;
  uint32_t first_block[kMaxSegments + 1];  // first_block[n] = workgroups of the launch
  uint32_t kind[kMaxSegments];             // KERNEL_STREAM_* (without the nullmap offset)
};

constexpr uint32_t kBlockThreads = 256;
constexpr uint32_t kStreamRingBytes = 16 * 1024;  // LDS ring of the streamed root step (4 batches of 1024 words)
// The stream kernels keep the ring at LDS offset 0 (ring addresses are then a plain mask), one
// mirror word behind it, then 32 doubles (the Pseudodecimal kernel's powers of ten by stream value:
// at a fixed address, so that a lookup is one shift and a read with an immediate offset) and the
// program copy after that.
constexpr uint32_t kStreamFracBase = kStreamRingBytes + 16;
constexpr uint32_t kStreamProgBase = kStreamFracBase + 32 * 8;
// LDS scratch of one header-chain walker: a table of (1024 + 256) 16-bit jump pointers.
constexpr uint32_t kWalkScratchBytes = (1024 + 256) * 2;
// SIMDFastPFor page descriptor size and the LDS work area of S_PFOR_DECODE:
//   [0, 4096)          a 16-byte entry per 256-value block of a page (256 blocks)
//   [4096, 8192)       the page's byte container when it is at most 4 KiB
//   [8192, 9472)       VariableByte tail (255 values of at most 5 bytes)
//   [9472, 25856)      a 4 KiB ring of packed block data per wave of the generic kernels; before
//                      the blocks are unpacked the same bytes hold the jump pointers of the
//                      byte-container chain, u16[4096 + 512], and the record offset per block, u16[256]
//   [25856, 29952)     high bits (value << b) of the exceptions of the current sub-range of blocks, in
//                      exception order (kPforExcCap of them; a sub-range with more is patched in memory)
//   [29952, 34048)     a 256-bit mask of exception positions per block of the sub-range
//   [34048, 35072)     per block, the exceptions in front of each 32-bit word of its mask (8 bytes)
// Sub-ranges of 128 blocks (two exception preparations per page): with 64 the preparations' barriers
// and lookup chains cost 0.38 ms per 1 GiB column instead of 0.35, with 256 the tables leave room for
// three workgroups per CU only (0.43 ms).
constexpr uint32_t kPforPageWords = 5 + 31 + 31;
constexpr uint32_t kPforSubBlocks = 128;  // blocks unpacked between two exception preparations
constexpr uint32_t kPforExcCap = 8 * kPforSubBlocks;
constexpr uint32_t kPforWorkBaseBytes = 25856;  // without the three exception-merge tables (step flag PFOR_NO_MERGE)
constexpr uint32_t kPforWorkBytes = kPforWorkBaseBytes + kPforExcCap * 4 + kPforSubBlocks * 32 + kPforSubBlocks * 8;
// Bits of S_PFOR_DECODE's a3 above the page count.
enum : uint32_t { PFOR_DELTA_BIT = 1u << 31, PFOR_NO_MERGE = 1u << 30, PFOR_PAGES_MASK = (1u << 30) - 1 };
// Streams with more groups than this get jump tables; shorter ones are walked without.
constexpr uint32_t kWalkScratchMinGroups = 64;
constexpr uint32_t kLdsCapBytes = 160 * 1024;  // all of a gfx950 CU's LDS; typical chunks need 5-40 KiB

}  // namespace btrplan
