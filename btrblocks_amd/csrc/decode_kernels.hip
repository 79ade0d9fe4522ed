// HIP decode kernels for gfx950 (MI355X): one workgroup per compressed column chunk, interpreting
// the flat program compiled by plan_builder.cpp (plan.h).
//
// What it replaces in the reference: the recursive virtual decompress() cascade
// (btrblocks/scheme/CompressionScheme.hpp:37-41, 70-74) with its scheme bodies
//   FastPFor FastBinaryPacking<32>+VariableByte decodeArray (via btrblocks/extern/FastPFOR.cpp:47-53)
//   FastPFor SIMDFastPFor<8>+VariableByte decodeArray (via btrblocks/extern/FastPFOR.cpp:20-31),
//   integer/PBP.cpp:37-49, 92-109,
//   templated/DynamicDictionary.hpp:108-216, templated/RLE.hpp:133-250,
//   templated/Frequency.hpp:99-136, templated/FixedDictionary.hpp:56-71,
//   integer/FOR.cpp:46-65, integer/OneValue.cpp:30-39, double/Pseudodecimal.cpp:348-452,
//   double/DoubleBP.cpp:41-57, extern/RoaringBitmap.cpp:32-114 (nullmap expansion).
//
// Two families of kernels (HBM-bound integer work, no MFMA):
//  * generic kernels (256 threads) interpret any program:
//    1. the program is copied into LDS;
//    2. every bit-packed stream gets a group index {word offset, width header} in LDS: the
//       FastBinaryPacking layout interleaves one header word per 128 values with the payload, so
//       header positions form a chain; one wave per stream resolves it batch by batch (walk_chain),
//       different streams on different waves;
//    3. short child streams (run lengths/values, exception lists, patches) are materialized into
//       LDS (or an HBM scratch slice), start-flag / exception bitmaps get a popcount rank table;
//    4. the root pipe is evaluated row by row - consecutive rows on consecutive lanes, several
//       rows per lane for table-driven bases - with full-length children (codes, deltas, digits,
//       exponents) bit-unpacked on the fly; nothing but the decoded column is written to HBM.
//  * stream kernels (512 threads) decode roots that read one full-length bit-packed stream: the
//    payload goes through an LDS ring exactly once (ring_pass), indexed and unpacked on the way.
// The evaluation loops are instantiated per base kind and post-op signature so that each loop only
// keeps the state of its own scheme in registers; everything is force-inlined.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "btr_format.h"
#include <type_traits>

#include "plan.h"

using namespace btrplan;

namespace {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

extern __shared__ __attribute__((aligned(16))) u8 lds_raw[];

// Pseudodecimal powers of ten, bit-identical to btrblocks/scheme/double/Pseudodecimal.cpp:19-43.
__constant__ double kFractions[23] = {
    1.0,
    0.1,
    0.01,
    0.001,
    0.0001,
    0.00001,
    0.000001,
    0.0000001,
    0.00000001,
    0.000000001,
    0.0000000001,
    0.00000000001,
    0.000000000001,
    0.0000000000001,
    0.00000000000001,
    0.000000000000001,
    0.0000000000000001,
    0.00000000000000001,
    0.000000000000000001,
    0.0000000000000000001,
    0.00000000000000000001,
    0.000000000000000000001,
    0.0000000000000000000001,
};

// ---- small loads ---------------------------------------------------------------------------
__device__ __forceinline__ u32 ldg32(const u8* p) {  // any byte alignment
  u32 v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ u64 ldg64(const u8* p) {
  u64 v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ __forceinline__ u32 ldg16(const u8* p) {
  u16 v;
  __builtin_memcpy(&v, p, 2);
  return v;
}
__device__ __forceinline__ u32 lds32(u32 off) { return *reinterpret_cast<const u32*>(lds_raw + off); }
__device__ __forceinline__ u64 lds64(u32 off) { return *reinterpret_cast<const u64*>(lds_raw + off); }
__device__ __forceinline__ void lds_st32(u32 off, u32 v) { *reinterpret_cast<u32*>(lds_raw + off) = v; }
__device__ __forceinline__ void lds_st64(u32 off, u64 v) { *reinterpret_cast<u64*>(lds_raw + off) = v; }
// Values read from the program tables in LDS are the same in every lane, but the compiler must
// be told: readfirstlane moves them to scalar registers so that loops over them stay scalar.
__device__ __forceinline__ u32 uniform(u32 v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const u8* uniform_ptr(const u8* p) {
  const u64 v = reinterpret_cast<u64>(p);
  return reinterpret_cast<const u8*>(((u64)uniform((u32)(v >> 32)) << 32) | uniform((u32)v));
}

typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
// Raw buffer descriptor over `bytes` bytes at p: loads past the end return 0 instead of faulting,
// so payload reads need no clamping.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, u32 bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// ---- per-workgroup context (all wave-uniform) ------------------------------------------------
struct Ctx {
  const u8* src;
  u8* dst;
  u8* nullmap;
  u8* scratch;
  u32* status;
  u32 steps_off, pipes_off, seqs_off;  // LDS byte offsets of the program tables
  u32 table_base;                      // LDS byte offset of table space
  u32 red_off;                         // 64 bytes of LDS for block-wide reductions
  u32 consts_off;                      // LDS byte offset of the program's constant words
  u32 side;                            // the item runs beside a plan's main launch (ITEM_SIDE_LAUNCH): raised wave priority
  u32 debug;                           // WorkItem::debug_flags
  u64 debug_buf;                       // WorkItem::debug_buf
};

__device__ __forceinline__ const Step& step_at(const Ctx& c, u32 i) {
  return *reinterpret_cast<const Step*>(lds_raw + c.steps_off + i * sizeof(Step));
}
__device__ __forceinline__ const Pipe& pipe_at(const Ctx& c, u32 i) {
  return *reinterpret_cast<const Pipe*>(lds_raw + c.pipes_off + i * sizeof(Pipe));
}
__device__ __forceinline__ const Seq& seq_at(const Ctx& c, u32 i) {
  return *reinterpret_cast<const Seq*>(lds_raw + c.seqs_off + i * sizeof(Seq));
}
__device__ __forceinline__ void raise(const Ctx& c, u32 bits) { atomicOr(c.status, bits); }
// Back from a raised priority (header-chain walkers) to the work item's own.
__device__ __forceinline__ void base_priority(const Ctx& c) {
  if (c.side)
    __builtin_amdgcn_s_setprio(2);
  else
    __builtin_amdgcn_s_setprio(0);
}
// Diagnostics: timestamp `slot` of this work item (constant 100 MHz clock), thread 0 only.
#define BTR_STAMP(c, slot)                                                              \
  do {                                                                                  \
    if (((c).debug & 8) && threadIdx.x == 0 && (slot) < kDebugSlots)                    \
      reinterpret_cast<u32*>((c).debug_buf)[slot] = (u32)wall_clock64();                \
  } while (0)
__device__ __forceinline__ void stamp(const WorkItem& it, u32 slot) {
  if ((it.debug_flags & 8) && threadIdx.x == 0 && slot < kDebugSlots)
    reinterpret_cast<u32*>(it.debug_buf)[slot] = (u32)wall_clock64();
}
__device__ __forceinline__ u32 seq_lds(const Ctx& c, u32 seq_id) { return c.table_base + uniform(seq_at(c, seq_id).off); }

// LDS offset of a streamed root's jump-table scratch, or NONE when the plan gave it up.
__device__ __forceinline__ u32 jump_scratch(const Ctx& c, const Step& root) {
  const u32 seq = uniform(root.a[3]);
  return seq == NONE ? NONE : seq_lds(c, seq);
}

// A resolved sequence: LDS byte offset or global pointer.
struct Tab {
  const u8* g;  // global base (scratch, source bytes or output); unused for LDS
  u32 lds;      // absolute LDS byte offset
  u32 count;
  u32 loc;
  u32 esize;
};

__device__ __forceinline__ Tab resolve_tab(const Ctx& c, u32 seq_id) {
  Tab t;
  t.g = nullptr;
  t.lds = 0;
  t.count = 0;
  t.loc = LOC_LDS;
  t.esize = 4;
  if (seq_id == NONE) return t;
  const Seq& s = seq_at(c, seq_id);
  const u32 off = uniform(s.off);
  t.count = uniform(s.count);
  t.loc = uniform(s.loc);
  t.esize = uniform(s.esize);
  switch (t.loc) {
    case LOC_LDS: t.lds = c.table_base + off; break;
    case LOC_SCRATCH: t.g = c.scratch + off; break;
    case LOC_SRC: t.g = c.src + off; break;
    case LOC_DST: t.g = c.dst + off; break;
    default: t.g = c.nullmap + off; break;
  }
  return t;
}
__device__ __forceinline__ u32 tab_ld32(const Tab& t, u32 i) {
  if (t.loc == LOC_LDS) return lds32(t.lds + 4 * i);
  return ldg32(t.g + 4ull * i);
}
__device__ __forceinline__ u64 tab_ld64(const Tab& t, u32 i) {
  if (t.loc == LOC_LDS) return lds64(t.lds + 8 * i);
  return ldg64(t.g + 8ull * i);
}
__device__ __forceinline__ u32 tab_ld_small(const Tab& t, u32 i) {  // esize 1, 2 or 4
  if (t.esize == 4) return tab_ld32(t, i);
  if (t.loc == LOC_LDS)
    return t.esize == 1 ? lds_raw[t.lds + i] : *reinterpret_cast<const u16*>(lds_raw + t.lds + 2 * i);
  return t.esize == 1 ? (u32)t.g[i] : ldg16(t.g + 2ull * i);
}
__device__ __forceinline__ void tab_st32(const Tab& t, u32 i, u32 v) {
  if (t.loc == LOC_LDS)
    lds_st32(t.lds + 4 * i, v);
  else
    *reinterpret_cast<u32*>(const_cast<u8*>(t.g) + 4ull * i) = v;  // scratch and output are aligned
}
__device__ __forceinline__ void tab_st64(const Tab& t, u32 i, u64 v) {
  if (t.loc == LOC_LDS)
    lds_st64(t.lds + 8 * i, v);
  else
    *reinterpret_cast<u64*>(const_cast<u8*>(t.g) + 8ull * i) = v;
}

// ---- bitmap + rank ---------------------------------------------------------------------------
__device__ __forceinline__ void bitmap_or(u32 bits_off, u32 row) {
  atomicOr(reinterpret_cast<u32*>(lds_raw + bits_off + 4 * (row >> 5)), 1u << (row & 31));
}
// bits: u64 words in LDS; rank[w] = set bits in words [0, w).
struct RankView {
  u32 bits_off, rank_off;
};
// set bits strictly below row i, and whether row i itself is set
__device__ __forceinline__ u32 rank_lt(const RankView& r, u32 i, bool* member) {
  const u32 w = i >> 6, p = i & 63;
  const u64 word = lds64(r.bits_off + 8 * w);
  *member = (word >> p) & 1;
  return lds32(r.rank_off + 4 * w) + __popcll(word & ((1ull << p) - 1));
}
// set bits at or below row i
__device__ __forceinline__ u32 rank_le(const RankView& r, u32 i) {
  const u32 w = i >> 6, p = i & 63;
  const u64 word = lds64(r.bits_off + 8 * w);
  return lds32(r.rank_off + 4 * w) + __popcll(word & ((2ull << p) - 1));
}

// ---- bit-packed stream view --------------------------------------------------------------------
struct BpView {
  const u8* base;  // payload (4-byte aligned); word 0 = packed value count
  u32 bytes;       // payload length; buffer loads beyond it return 0
  u32 rounded;     // values packed by FastBinaryPacking (multiple of 32)
  u32 index_off;   // LDS offset of {word offset, header} per 128 values
  u32 tail_off;    // LDS offset of the decoded VariableByte tail
};

__device__ __forceinline__ BpView resolve_bp(const Ctx& c, const Pipe& p) {
  BpView b;
  b.base = c.src + uniform(p.a[0]);
  b.bytes = uniform(p.a[1]) * 4;
  b.rounded = uniform(p.a[3]);
  b.index_off = seq_lds(c, uniform(p.a[2]));
  const u32 tail = uniform(p.a[4]);
  b.tail_off = tail == NONE ? 0 : seq_lds(c, tail);
  return b;
}

// One value of a miniblock: `base_word` is the first word of the miniblock, value k occupies bits
// [k*b, (k+1)*b) of its little-endian bit stream. One 8-byte buffer load covers any value.
__device__ __forceinline__ u32 bp_extract(__amdgpu_buffer_rsrc_t rsrc, u32 base_word, u32 b, u32 k) {
  const u32 bit = k * b;
  const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (base_word + (bit >> 5)) * 4, 0, 0);
  const u32 v = __builtin_amdgcn_alignbit(w.y, w.x, bit);
  return b >= 32 ? v : (v & ((1u << b) - 1u));
}

// Value `i` of a FastBinaryPacking<32>+VariableByte stream. Group g = i/128 starts at word
// index[g].x with header index[g].y = (b0<<24)|(b1<<16)|(b2<<8)|b3; miniblock m = (i/32)%4 starts
// after the header and the b_j words of the miniblocks before it.
// ALIGNED: the wave's lanes evaluate 64 consecutive rows starting at a multiple of 64, so the
// group is wave-uniform and its index entry is fetched once.
template <bool ALIGNED>
__device__ __forceinline__ u32 bp_get(const BpView& e, u32 i) {
  if (i >= e.rounded) return lds32(e.tail_off + 4 * ((i - e.rounded) & 31));
  u32 off, hdr;
  if (ALIGNED) {
    const u32 g = uniform(i >> 7);
    const u64 ih = lds64(e.index_off + 8 * g);
    off = uniform((u32)ih);
    hdr = uniform((u32)(ih >> 32));
  } else {
    const u64 ih = lds64(e.index_off + 8 * (i >> 7));
    off = (u32)ih;
    hdr = (u32)(ih >> 32);
  }
  const u32 m = (i >> 5) & 3, k = i & 31;
  u32 b = (hdr >> (24 - 8 * m)) & 0xFF;
  b = b > 32 ? 32 : b;
  const u32 before = __builtin_amdgcn_sad_u8((u32)((u64)hdr >> (32 - 8 * m)), 0u, 0u);
  return bp_extract(make_rsrc(e.base, e.bytes), off + 1 + before, b, k);
}

// ---- resolved integer pipe -------------------------------------------------------------------
constexpr u32 GEN = 255;  // template argument "base kind decided at run time"

struct IntEval {
  u32 base;
  u32 count;
  u32 k0;       // B_ONE value / B_FREQ top value
  BpView bp;    // B_BP
  RankView rk;  // B_RLE / B_FREQ
  Tab tab;      // B_SEQ source / B_RLE run values / B_FREQ exceptions
  u32 n_post;
  u32 post_type[MAX_POST];
  u32 post_bias[MAX_POST];
  Tab post_tab[MAX_POST];
};

__device__ __forceinline__ IntEval resolve_int(const Ctx& c, u32 pipe_id) {
  const Pipe& p = pipe_at(c, pipe_id);
  IntEval e;
  e.base = uniform(p.base);
  e.count = uniform(p.count);
  e.k0 = 0;
  e.bp = BpView{nullptr, 0, 0, 0, 0};
  e.rk = RankView{0, 0};
  e.tab = resolve_tab(c, NONE);
  switch (e.base) {
    case B_SEQ: e.tab = resolve_tab(c, uniform(p.a[0])); break;
    case B_ONE: e.k0 = uniform(p.a[0]); break;
    case B_BP: e.bp = resolve_bp(c, p); break;
    case B_RLE:
      e.rk.bits_off = seq_lds(c, uniform(p.a[0]));
      e.rk.rank_off = seq_lds(c, uniform(p.a[1]));
      e.tab = resolve_tab(c, uniform(p.a[2]));
      break;
    case B_FREQ:
      e.k0 = uniform(p.a[0]);
      e.rk.bits_off = seq_lds(c, uniform(p.a[2]));
      e.rk.rank_off = seq_lds(c, uniform(p.a[3]));
      e.tab = resolve_tab(c, uniform(p.a[4]));
      break;
    default: break;
  }
  e.n_post = uniform(p.n_post);
#pragma unroll
  for (int k = 0; k < MAX_POST; k++) {
    const u32 type = k < (int)e.n_post ? uniform(p.post[k].type) : 0;
    const u32 arg = uniform(p.post[k].arg);
    e.post_type[k] = type;
    e.post_bias[k] = arg;
    e.post_tab[k] = resolve_tab(c, type == POST_DICT ? arg : NONE);
  }
  return e;
}

// Dictionary gathers and FOR bias adds queued behind the base value
// (templated/DynamicDictionary.hpp:125-158: dest[i] = dict[codes[i]]; integer/FOR.cpp:62-64).
// The hot loops are instantiated per post-op signature so that no per-value branching remains;
// a code outside the dictionary is clamped and remembered in `bad` (reported once per loop).
enum : u32 { PS_NONE = 0, PS_ADD = 1, PS_DICT_LDS = 2, PS_DICT_GLB = 3, PS_GEN = 4 };

__device__ __forceinline__ u32 post_signature(const IntEval& e) {
  if (e.n_post == 0) return PS_NONE;
  if (e.n_post == 1) {
    if (e.post_type[0] == POST_ADD) return PS_ADD;
    if (e.post_tab[0].count == 0) return PS_GEN;
    return e.post_tab[0].loc == LOC_LDS ? PS_DICT_LDS : PS_DICT_GLB;
  }
  return PS_GEN;
}

__device__ __forceinline__ u32 apply_posts_generic(const IntEval& e, u32 v, u32& bad) {
#pragma unroll
  for (int k = 0; k < MAX_POST; k++) {
    if (k < (int)e.n_post) {
      if (e.post_type[k] == POST_DICT) {
        const Tab& d = e.post_tab[k];
        const u32 n = d.count ? d.count : 1;
        bad |= v >= d.count;
        v = v < n ? v : n - 1;
        v = d.count ? tab_ld32(d, v) : 0;
      } else {
        v += e.post_bias[k];
      }
    }
  }
  return v;
}

template <u32 PS>
__device__ __forceinline__ u32 apply_posts(const IntEval& e, u32 v, u32& bad) {
  if (PS == PS_NONE) return v;
  if (PS == PS_ADD) return v + e.post_bias[0];
  if (PS == PS_DICT_LDS || PS == PS_DICT_GLB) {
    const u32 n = e.post_tab[0].count;
    bad |= v >= n;
    v = v < n ? v : n - 1;
    return PS == PS_DICT_LDS ? lds32(e.post_tab[0].lds + 4 * v) : ldg32(e.post_tab[0].g + 4ull * v);
  }
  return apply_posts_generic(e, v, bad);
}

template <u32 BASE, bool ALIGNED>
__device__ __forceinline__ u32 eval_int_base(const IntEval& e, u32 i) {
  if (BASE == B_SEQ) return tab_ld_small(e.tab, i);
  if (BASE == B_ONE) return e.k0;
  if (BASE == B_BP) return bp_get<ALIGNED>(e.bp, i);
  if (BASE == B_RLE) {
    // run index = (run starts at or before row i) - 1   (templated/RLE.hpp:164-191)
    u32 r = rank_le(e.rk, i);
    r = r ? r - 1 : 0;
    r = r < e.tab.count ? r : e.tab.count - 1;
    return tab_ld32(e.tab, r);
  }
  if (BASE == B_FREQ) {
    // top value unless the row is a bitmap member; the k-th member takes exceptions[k]
    // (templated/Frequency.hpp:118-135)
    bool member;
    u32 r = rank_lt(e.rk, i, &member);
    u32 v = e.k0;
    if (member) {
      r = r < e.tab.count ? r : e.tab.count - 1;
      v = tab_ld32(e.tab, r);
    }
    return v;
  }
  return 0;
}

template <u32 BASE, bool ALIGNED, u32 PS = PS_GEN>
__device__ __forceinline__ u32 eval_int(const IntEval& e, u32 i, u32& bad) {
  u32 v;
  if (BASE != GEN) {
    v = eval_int_base<BASE, ALIGNED>(e, i);
  } else {
    switch (e.base) {
      case B_SEQ: v = eval_int_base<B_SEQ, ALIGNED>(e, i); break;
      case B_ONE: v = eval_int_base<B_ONE, ALIGNED>(e, i); break;
      case B_BP: v = eval_int_base<B_BP, ALIGNED>(e, i); break;
      case B_RLE: v = eval_int_base<B_RLE, ALIGNED>(e, i); break;
      case B_FREQ: v = eval_int_base<B_FREQ, ALIGNED>(e, i); break;
      default: v = 0; break;
    }
  }
  return apply_posts<PS>(e, v, bad);
}

// ---- resolved double pipe --------------------------------------------------------------------
struct DblEval {
  u32 base;
  u32 count;
  u32 k0, k1;   // B_ONE / B_FREQ value bits; B_PSEUDO: k0 = converted_count
  RankView rk;
  Tab tab;      // B_SEQ / B_RLE values / B_FREQ exceptions / B_DICTD dictionary / B_PSEUDO patches
  IntEval ia;   // B_DICTD codes / B_PSEUDO digits / B_DBP word stream
  IntEval ib;   // B_PSEUDO exponents
  u32 frac_lo, frac_hi;  // B_PSEUDO: lane l holds 10^-min(l, 22), fetched per row by a lane permute
};

__device__ __forceinline__ DblEval resolve_dbl(const Ctx& c, u32 pipe_id) {
  const Pipe& p = pipe_at(c, pipe_id);
  DblEval e;
  e.base = uniform(p.base);
  e.count = uniform(p.count);
  e.k0 = e.k1 = 0;
  e.rk = RankView{0, 0};
  e.tab = resolve_tab(c, NONE);
  e.ia.base = e.ib.base = B_ONE;
  e.ia.n_post = e.ib.n_post = 0;
  e.frac_lo = e.frac_hi = 0;
  const u32 a0 = uniform(p.a[0]), a1 = uniform(p.a[1]), a2 = uniform(p.a[2]), a3 = uniform(p.a[3]),
            a4 = uniform(p.a[4]), a5 = uniform(p.a[5]);
  switch (e.base) {
    case B_SEQ: e.tab = resolve_tab(c, a0); break;
    case B_ONE:
      e.k0 = a0;
      e.k1 = a1;
      break;
    case B_RLE:
      e.rk.bits_off = seq_lds(c, a0);
      e.rk.rank_off = seq_lds(c, a1);
      e.tab = resolve_tab(c, a2);
      break;
    case B_FREQ:
      e.k0 = a0;
      e.k1 = a1;
      e.rk.bits_off = seq_lds(c, a2);
      e.rk.rank_off = seq_lds(c, a3);
      e.tab = resolve_tab(c, a4);
      break;
    case B_DICTD:
      e.ia = resolve_int(c, a0);
      e.tab = resolve_tab(c, a1);
      break;
    case B_PSEUDO:
      if (a0 != NONE) e.ia = resolve_int(c, a0);
      e.ib = resolve_int(c, a1);
      e.tab = resolve_tab(c, a2);
      e.rk.bits_off = seq_lds(c, a3);
      e.rk.rank_off = seq_lds(c, a4);
      e.k0 = a5;
      {
        const u32 l = threadIdx.x & 63;
        const u64 f = (u64)__double_as_longlong(kFractions[l < btrfmt::DEC_MAX_EXPONENT ? l : btrfmt::DEC_MAX_EXPONENT]);
        e.frac_lo = (u32)f, e.frac_hi = (u32)(f >> 32);
      }
      break;
    case B_DBP:
      e.ia.base = B_BP;
      e.ia.bp = resolve_bp(c, p);
      break;
    default: break;
  }
  return e;
}

// IA / IB: base kinds of the nested integer pipes when known at compile time, else GEN.
// IPS: post-op signature of the nested integer pipes (PS_NONE when both are bare).
template <u32 BASE, u32 IA, u32 IB, u32 IPS = PS_GEN>
__device__ __forceinline__ u64 eval_dbl(const DblEval& e, u32 i, u32& bad) {
  if (BASE == B_SEQ) return tab_ld64(e.tab, i);
  if (BASE == B_ONE) return ((u64)e.k1 << 32) | e.k0;
  if (BASE == B_RLE) {
    u32 r = rank_le(e.rk, i);
    r = r ? r - 1 : 0;
    r = r < e.tab.count ? r : e.tab.count - 1;
    return tab_ld64(e.tab, r);
  }
  if (BASE == B_FREQ) {
    bool member;
    u32 r = rank_lt(e.rk, i, &member);
    u64 v = ((u64)e.k1 << 32) | e.k0;
    if (member) {
      r = r < e.tab.count ? r : e.tab.count - 1;
      v = tab_ld64(e.tab, r);
    }
    return v;
  }
  if (BASE == B_DICTD) {
    u32 code = eval_int<IA, true, IPS>(e.ia, i, bad);
    const u32 n = e.tab.count ? e.tab.count : 1;
    bad |= code >= e.tab.count;
    code = code < n ? code : n - 1;
    return e.tab.count ? tab_ld64(e.tab, code) : 0;
  }
  if (BASE == B_PSEUDO) {
    // double/Pseudodecimal.cpp:424-436: exponent 23 takes the next patch, anything else is
    // double(int32 digits) * 10^-exponent - one exact conversion and one IEEE multiply.
    const u32 ex = eval_int<IB, true, IPS>(e.ib, i, bad);
    bool member;
    const u32 xr = rank_lt(e.rk, i, &member);
    u32 di = i - xr;
    di = di < e.k0 ? di : (e.k0 ? e.k0 - 1 : 0);
    const int digits = e.k0 ? (int)eval_int<IA, false, IPS>(e.ia, di, bad) : 0;
    u32 idx = ex & btrfmt::DEC_INDEX_MASK;
    bad |= (ex != btrfmt::DEC_EXCEPTION_CODE && idx > btrfmt::DEC_MAX_EXPONENT) ? 2u : 0u;
    idx = idx > btrfmt::DEC_MAX_EXPONENT ? btrfmt::DEC_MAX_EXPONENT : idx;
    const u64 fbits = ((u64)(u32)__builtin_amdgcn_ds_bpermute(idx << 2, e.frac_hi) << 32) |
                      (u32)__builtin_amdgcn_ds_bpermute(idx << 2, e.frac_lo);
    u64 r = (u64)__double_as_longlong((double)digits * __longlong_as_double((long long)fbits));
    // the patch candidate is loaded for every row: a load inside the rare divergent branch would
    // expose its whole memory latency whenever any lane of the wave takes it
    const u32 pi = xr < e.tab.count ? xr : e.tab.count - 1;
    const u64 patch = e.tab.count ? tab_ld64(e.tab, pi) : 0;
    if (ex == btrfmt::DEC_EXCEPTION_CODE) r = patch;
    return r;
  }
  if (BASE == B_DBP) {
    const u32 lo = bp_get<false>(e.ia.bp, 2 * i), hi = bp_get<false>(e.ia.bp, 2 * i + 1);
    return ((u64)hi << 32) | lo;
  }
  return 0;
}

// ---- block-wide exclusive scan of one value per thread ------------------------------------------
template <u32 THREADS = kBlockThreads>
__device__ __forceinline__ u32 block_exclusive_scan(const Ctx& c, u32 v, u32* total) {
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 t = __shfl_up(incl, d, 64);
    if (lane >= (u32)d) incl += t;
  }
  __syncthreads();  // previous users of the reduction scratch are done
  if (lane == 63) lds_st32(c.red_off + 4 * wave, incl);
  __syncthreads();
  u32 before = 0, sum = 0;
#pragma unroll
  for (u32 w = 0; w < THREADS / 64; w++) {
    const u32 s = lds32(c.red_off + 4 * w);
    before += w < wave ? s : 0;
    sum += s;
  }
  *total = sum;
  return before + incl - v;
}

// ---- step: bit-unpack group index ---------------------------------------------------------------
// One wave per stream. The payload is taken in batches of 1024 words, sixteen per lane (four
// coalesced 16-byte buffer loads), the next batch in flight while the current one is walked.
// Every word is turned into "group length if this word were a header" (1 + the sum of its four
// width bytes); the chain of real headers is then followed through the batch (see the loop).
// Group start positions go to the LDS index; header words are gathered in parallel afterwards.
__device__ __forceinline__ u32 group_lengths(const u32x4& w) {
  auto glen = [](u32 x) {
    const u32 s = 1 + __builtin_amdgcn_sad_u8(x, 0u, 0u);
    return s > 255 ? 255u : s;
  };
  return glen(w.x) | (glen(w.y) << 8) | (glen(w.z) << 16) | (glen(w.w) << 24);
}

constexpr u32 kWalkBatch = 4;  // 256-word windows per batch (1024 words, 16 per lane)
constexpr u32 kBatchWords = 256 * kWalkBatch;

// "Group length if the word at batch position p were a header", gathered from the lane that
// holds it: lane (p/4)%64 keeps byte p%4 of window p/256.
__device__ __forceinline__ u32 length_at(const u32 (&d)[kWalkBatch], u32 p) {
  const u32 src_lane = ((p >> 2) & 63) << 2;
  u32 v = __builtin_amdgcn_ds_bpermute(src_lane, d[0]);
#pragma unroll
  for (u32 k = 1; k < kWalkBatch; k++) {
    const u32 t = __builtin_amdgcn_ds_bpermute(src_lane, d[k]);
    v = (p >> 8) == k ? t : v;
  }
  return (v >> (8 * (p & 3))) & 0xFF;
}

// The same for a wave-uniform position: scalar lane reads, no cross-lane traffic.
__device__ __forceinline__ u32 length_at_uniform(const u32 (&d)[kWalkBatch], u32 p) {
  const u32 sl = (p >> 2) & 63;
  u32 v = __builtin_amdgcn_readlane(d[0], sl);
#pragma unroll
  for (u32 k = 1; k < kWalkBatch; k++) {
    const u32 t = __builtin_amdgcn_readlane(d[k], sl);
    v = (p >> 8) == k ? t : v;
  }
  return (v >> (8 * (p & 3))) & 0xFF;
}

// Follows the header chain "next header = this header + 1 + sum of its four width bytes" through
// one batch whose per-word lengths sit in d[]. Two kinds of step:
//   run      the length D at `pos` is read and lane j checks the word at pos + j*D: the leading run
//            of lanes that also find length D are real headers (lane 0 is one by construction, and
//            a confirmed lane j makes the guess of lane j+1 exact), so a stretch of equally wide
//            groups costs one step for up to 64 groups;
//   jumps    when widths change from group to group the chain is resolved by pointer doubling in
//            an LDS table: J_0[p] = p + length(p) for all 1024 positions of the batch, J_{r+1} =
//            J_r o J_r; the set of reached positions doubles per round (lane j ends up with the
//            j-th group), so a batch costs O(log groups) wave-wide rounds instead of one serial
//            step per group.
// The jump tables need kWalkScratchBytes of LDS (`jump_off`, or NONE: run steps only - enough for
// short streams). emit(first_group, count, position) is called with a per-lane position for
// lanes < count.
constexpr u32 kJumpEntries = kBatchWords + 256;  // positions past the batch map to themselves
static_assert(kWalkScratchBytes == kJumpEntries * 2, "one u16 table; the size is shared with the plan compiler");

struct ChainState {
  u32 pos, g;   // next header position, groups found
  u32 spec_ok;  // the last batch was made of equally long groups: try a run step first
};

__device__ __forceinline__ u32 lds16(u32 off) { return *reinterpret_cast<const u16*>(lds_raw + off); }

// Entries [kBatchWords, kJumpEntries) of the table: self loops. Once per walker scratch.
__device__ __forceinline__ void jump_tables_init(u32 jump_off) {
  if (jump_off == NONE) return;
  const u32 lane = threadIdx.x & 63;
  const u32 p = kBatchWords + 4 * lane;
  lds_st64(jump_off + 2 * p, (u64)(p | ((p + 1) << 16)) | ((u64)((p + 2) | ((p + 3) << 16)) << 32));
}

template <typename Emit>
__device__ __forceinline__ void walk_chain(const u32 (&d)[kWalkBatch], u32 bb, u32 groups, u32 jump_off, ChainState& st,
                                           Emit emit) {
  const u32 lane = threadIdx.x & 63;
  while (st.pos < bb + kBatchWords && st.g < groups) {
    const u32 rel = st.pos - bb;
    if (st.spec_ok || jump_off == NONE) {
      const u32 D = length_at_uniform(d, rel);
      const u32 pj = rel + lane * D;
      const bool inside = pj < kBatchWords;
      const u32 dj = length_at(d, inside ? pj : rel);
      const u64 ok = __ballot(inside && dj == D);
      u32 m = (u32)__builtin_ctzll(~ok | (1ull << 63));
      m = m == 63 && (ok >> 63) ? 64 : m;
      m = m < groups - st.g ? m : groups - st.g;
      emit(st.g, m, bb + pj);
      st.g = uniform(st.g + m);
      st.pos = uniform(st.pos + m * D);
      if (m < 64 && st.pos < bb + kBatchWords) st.spec_ok = 0;  // the run broke inside the batch
      continue;
    }
    // ---- pointer doubling from `rel`
#pragma unroll
    for (u32 k = 0; k < kWalkBatch; k++) {  // J_0 of this lane's own positions 256k + 4*lane + i
      const u32 p = 256 * k + 4 * lane;
      const u32 j0 = p + (d[k] & 0xFF), j1 = p + 1 + ((d[k] >> 8) & 0xFF), j2 = p + 2 + ((d[k] >> 16) & 0xFF),
                j3 = p + 3 + (d[k] >> 24);
      lds_st64(jump_off + 2 * p, (u64)(j0 | (j1 << 16)) | ((u64)(j2 | (j3 << 16)) << 32));
    }
    u32 node = rel, cnt = 1;
    for (;;) {
      const u32 nxt = lds16(jump_off + 2 * node);
      const u32 pulled = __builtin_amdgcn_ds_bpermute((lane - cnt) << 2, nxt);
      node = (lane >= cnt && lane < 2 * cnt) ? pulled : node;
      cnt *= 2;
      const u32 far = __builtin_amdgcn_readlane(node, cnt - 1);
      if (far >= kBatchWords || cnt == 64) break;
      // J_{r+1} = J_r o J_r in place: the wave reads all of J_r (its LDS reads are served in
      // program order) before it writes any of J_{r+1}
      u32 lo[kWalkBatch], hi[kWalkBatch];
#pragma unroll
      for (u32 k = 0; k < kWalkBatch; k++) {
        const u64 own = lds64(jump_off + 2 * (256 * k + 4 * lane));
        const u32 j0 = lds16(jump_off + 2 * ((u32)own & 0xFFFF)), j1 = lds16(jump_off + 2 * ((u32)own >> 16)),
                  j2 = lds16(jump_off + 2 * ((u32)(own >> 32) & 0xFFFF)), j3 = lds16(jump_off + 2 * (u32)(own >> 48));
        lo[k] = j0 | (j1 << 16);
        hi[k] = j2 | (j3 << 16);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (u32 k = 0; k < kWalkBatch; k++) lds_st64(jump_off + 2 * (256 * k + 4 * lane), (u64)lo[k] | ((u64)hi[k] << 32));
    }
    u32 m = (u32)__popcll(__ballot(lane < cnt && node < kBatchWords));
    m = m < groups - st.g ? m : groups - st.g;
    emit(st.g, m, bb + node);
    // equally spaced groups: the next batch may start with a run step again
    const u32 step0 = __builtin_amdgcn_readlane(node, 1) - rel;
    const u32 prev = __builtin_amdgcn_ds_bpermute(((lane ? lane : 1) - 1) << 2, node);
    st.spec_ok = m >= 4 && __ballot(lane >= 1 && lane < m && node - prev != step0) == 0;
    u32 next;
    if (m < cnt) {
      next = __builtin_amdgcn_readlane(node, m);
      st.g = uniform(st.g + m);
    } else if (cnt == 64) {
      // All 64 lanes hold groups inside the batch (narrow groups) and the table holds J^32: the next
      // 64 groups are J^32(J^32(node)), two gathers, instead of a new table and six more rounds.
      // Positions behind the batch map to themselves, so a lane whose chain has left the batch keeps
      // its exit position.
      st.g = uniform(st.g + m);
      for (;;) {
        const u32 node2 = lds16(jump_off + 2 * lds16(jump_off + 2 * node));
        u32 m2 = (u32)__popcll(__ballot(node2 < kBatchWords));
        m2 = m2 < groups - st.g ? m2 : groups - st.g;
        emit(st.g, m2, bb + node2);
        st.g = uniform(st.g + m2);
        if (m2 < 64) {
          next = __builtin_amdgcn_readlane(node2, m2);
          break;
        }
        node = node2;
      }
      st.spec_ok = 0;
    } else {  // fewer than 64 lanes, all inside the batch, and the chain goes on: step behind the last one
      const u32 last = __builtin_amdgcn_readlane(node, cnt - 1);
      next = last + length_at_uniform(d, last);
      st.g = uniform(st.g + m);
    }
    st.pos = uniform(bb + next);
  }
}

// Walk state, kept in the registers of the walking wave.
struct Walker {
  const u8* base;
  u32 W, rounded, groups, index_off;
  u32 jump_off;    // LDS scratch for the jump tables, or NONE
  u32 pos, g, bb;  // next header position, groups found, first word of the current batch
  u32 bad;
  u32 tail_word;          // word max(W, 64) - 64 + lane of the stream: covers any VariableByte tail
  u32 d[kWalkBatch];      // packed group lengths of the current batch
  u32x4 nxt[kWalkBatch];  // next batch, in flight
};

__device__ __forceinline__ u32x4 load_window(__amdgpu_buffer_rsrc_t rsrc, u32 first_word, u32 lane) {
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (first_word + 4 * lane) * 4, 0, 0);  // past the end: 0
}

// The last 64 words of a stream, one per lane (lane l: word max(W, 64) - 64 + l). Loaded before
// the walk so that the tail decode at its end does not wait for memory.
__device__ __forceinline__ u32 load_tail_words(__amdgpu_buffer_rsrc_t rsrc, u32 W) {
  const u32 first = W > 64 ? W - 64 : 0;
  return __builtin_amdgcn_raw_buffer_load_b32(rsrc, (first + (threadIdx.x & 63)) * 4, 0, 0);
}

__device__ __forceinline__ void walker_init(Walker& w, const Ctx& c, const Pipe& p, u32 jump_seq) {
  const u32 lane = threadIdx.x & 63;
  w.jump_off = jump_seq == NONE ? NONE : seq_lds(c, jump_seq);
  w.base = c.src + uniform(p.a[0]);
  w.W = uniform(p.a[1]);
  w.rounded = uniform(p.a[3]);
  w.groups = uniform(p.a[5]);
  w.index_off = seq_lds(c, uniform(p.a[2]));
  const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(w.base, w.W * 4);
  u32x4 cur[kWalkBatch];
#pragma unroll
  for (u32 k = 0; k < kWalkBatch; k++) cur[k] = load_window(rsrc, 256 * k, lane);
#pragma unroll
  for (u32 k = 0; k < kWalkBatch; k++) w.nxt[k] = load_window(rsrc, kBatchWords + 256 * k, lane);
  w.tail_word = load_tail_words(rsrc, w.W);
#pragma unroll
  for (u32 k = 0; k < kWalkBatch; k++) w.d[k] = group_lengths(cur[k]);
  w.pos = 1;
  w.g = w.bb = w.bad = 0;
}

// Finds the start of every group (or stops where the stream ends), then completes the index
// entries {word offset, header}.
__device__ __forceinline__ void walker_run(Walker& w) {
  const u32 lane = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(w.base, w.W * 4);
  ChainState st{w.pos, w.g, 1};
  const u32 index_off = w.index_off;
  jump_tables_init(w.jump_off);
  for (;;) {
    walk_chain(w.d, w.bb, w.groups, w.jump_off, st, [&](u32 first, u32 count, u32 position) {
      if (lane < count) lds_st32(index_off + 8 * (first + lane), position);
    });
    if (st.g >= w.groups || w.bb + kBatchWords >= w.W) break;
    w.bb += kBatchWords;  // move to the next batch
#pragma unroll
    for (u32 k = 0; k < kWalkBatch; k++) w.d[k] = group_lengths(w.nxt[k]);
#pragma unroll
    for (u32 k = 0; k < kWalkBatch; k++) w.nxt[k] = load_window(rsrc, w.bb + kBatchWords + 256 * k, lane);
  }
  w.pos = st.pos;
  w.g = st.g;
  // fetch and validate the header word of every group (every width byte must be <= 32), four
  // loads in flight per lane
  for (u32 g0 = 0; g0 < w.g; g0 += 256) {
    u32 hv[4];
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
      const u32 gg = g0 + 64 * j + lane;
      const u32 gp = gg < w.g ? lds32(w.index_off + 8 * gg) : 0x3FFFFFFFu;  // past the end: reads 0
      hv[j] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, gp * 4, 0, 0);
    }
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
      const u32 gg = g0 + 64 * j + lane;
      if ((hv[j] & 0xC0C0C0C0u) || ((hv[j] + 0x5F5F5F5Fu) & 0x80808080u)) w.bad = 1;
      if (gg < w.g) lds_st32(w.index_off + 8 * gg + 4, hv[j]);
    }
  }
}

// After the last group: flags, the end-of-packing sentinel and the VariableByte tail (fewer
// than 32 values: 7 bits per byte, low group first, a set top bit ends a value).
struct WalkEnd {
  u32 W, rounded, groups, index_off, pos, g, bad;
  u32 tail_word;  // load_tail_words()
};
__device__ __forceinline__ void walker_finish(const WalkEnd& w, const Ctx& c, const Pipe& p) {
  const u32 lane = threadIdx.x & 63;
  u32 bad = w.bad | (w.g < w.groups);  // the stream ended before all groups were found
  u32 pos = w.pos;
  if (pos > w.W) {
    bad = 1;
    pos = w.W;
  }
  for (u32 gg = w.g + lane; gg < w.groups; gg += 64) lds_st64(w.index_off + 8 * gg, 0);
  if (lane == 0) lds_st64(w.index_off + 8 * w.groups, (u64)pos);
  if (bad) raise(c, ST_BP_HEADER);
  const u32 tail_seq = uniform(p.a[4]);
  if (tail_seq != NONE) {
    const u32 tail_off = seq_lds(c, tail_seq);
    const u32 count = uniform(p.count);
    const u32 stream_values = uniform(p.base) == B_DBP ? 2 * count : count;
    const u32 expect = stream_values - w.rounded;
    // a tail of more than 64 words cannot be one (31 values of at most 5 bytes): its first 64
    // words are skipped and the count check below reports it
    const u32 first = w.W > 64 ? w.W - 64 : 0;
    const u32 skip = pos > first ? pos - first : 0;  // lanes in front of the tail
    const u32 tail_words = w.W - first - skip;
    // Lane l holds word first + l; the tail is lanes [skip, skip + tail_words). A byte with its top
    // bit set ends a value. All values are decoded at once: ballots over the four byte positions
    // give every terminator its value index, and a value (at most five bytes) lies inside the
    // 8-byte window {previous lane's word, own word}. Words outside the tail count as all
    // terminators, which bounds the first value on the left.
    const bool mine = lane >= skip && lane < skip + tail_words;
    const u32 cur = mine ? w.tail_word : 0x80808080u;
    const u32 left = __builtin_amdgcn_ds_bpermute(((lane ? lane : 1) - 1) << 2, cur);
    const u64 win = ((u64)cur << 32) | (lane ? left : 0x80808080u);
    u32 idx = 0, produced = 0;
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
      const u64 t = __builtin_amdgcn_ballot_w64(mine && ((cur >> (8 * j + 7)) & 1));
      idx += __builtin_amdgcn_mbcnt_hi((u32)(t >> 32), __builtin_amdgcn_mbcnt_lo((u32)t, 0u));  // in the lanes below
      produced += (u32)__popcll(t);
    }
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
      if (mine && ((cur >> (8 * j + 7)) & 1)) {
        const u32 p8 = 4 + j;  // the terminator's byte inside the window
        const u32 c1 = ((u32)(win >> (8 * (p8 - 1) + 7)) & 1) ^ 1;
        const u32 c2 = c1 & (((u32)(win >> (8 * (p8 - 2) + 7)) & 1) ^ 1);
        const u32 c3 = c2 & (((u32)(win >> (8 * (p8 - 3) + 7)) & 1) ^ 1);
        const u32 c4 = c3 & (((u32)(win >> (8 * (p8 - 4) + 7)) & 1) ^ 1);
        const u32 len = 1 + c1 + c2 + c3 + c4;
        const u64 f = (win >> (8 * (p8 + 1 - len))) & ((1ull << (8 * len)) - 1);
        const u32 v = ((u32)f & 0x7F) | (((u32)(f >> 8) & 0x7F) << 7) | (((u32)(f >> 16) & 0x7F) << 14) |
                      (((u32)(f >> 24) & 0x7F) << 21) | (((u32)(f >> 32) & 0x7F) << 28);
        if (idx < 32) lds_st32(tail_off + 4 * idx, v);
        idx++;
      }
    }
    if ((produced != expect || pos < first) && lane == 0) raise(c, ST_BP_TAIL);
  }
}

__device__ __forceinline__ void bp_index_wave(const Ctx& c, const Step& s) {
  const Pipe& p = pipe_at(c, uniform(s.a[0]));
  Walker w;
  __builtin_amdgcn_s_setprio(3);  // a serial chain other waves wait for
  walker_init(w, c, p, uniform(s.a[1]));
  walker_run(w);
  base_priority(c);
  walker_finish(WalkEnd{w.W, w.rounded, w.groups, w.index_off, w.pos, w.g, w.bad, w.tail_word}, c, p);
}

// ---- step: materialize a pipe into a sequence ----------------------------------------------------
constexpr u32 kUnroll = 4;

__device__ __forceinline__ void report(const Ctx& c, u32 bad) {
  if (bad & 1) raise(c, ST_DICT_CODE);
  if (bad & 2) raise(c, ST_EXPONENT);
}

template <u32 BASE, u32 PS>
__device__ __forceinline__ void mat_int(const Ctx& c, const IntEval& e, const Tab& out, u32 count) {
  const u32 tid = threadIdx.x;
  u32 bad = 0;
  for (u32 base = 0; base < count; base += kBlockThreads * kUnroll) {
    u32 v[kUnroll];
#pragma unroll
    for (u32 u = 0; u < kUnroll; u++) {
      const u32 i = base + u * kBlockThreads + tid;
      v[u] = eval_int<BASE, true, PS>(e, i < count ? i : count - 1, bad);
    }
#pragma unroll
    for (u32 u = 0; u < kUnroll; u++) {
      const u32 i = base + u * kBlockThreads + tid;
      if (i < count) tab_st32(out, i, v[u]);
    }
  }
  report(c, bad);
}

// ---- four consecutive rows per lane -----------------------------------------------------------
// Root steps of table-driven bases (fill, copy, run expansion, exception patching) write 16 bytes
// per lane: one bitmap word and one rank lookup serve four rows, and a wave stores 1 KiB per
// instruction.
// TAB_LDS: the value table is known to sit in LDS (no location test per load, so the four loads
// of a lane are in flight together).
template <u32 BASE, bool TAB_LDS = false>
__device__ __forceinline__ u32x4 eval_int4_base(const IntEval& e, u32 i, __amdgpu_buffer_rsrc_t seq_rsrc) {
  u32x4 v;
  if (BASE == B_ONE) {
    v.x = v.y = v.z = v.w = e.k0;
  } else if (BASE == B_SEQ) {
    v = __builtin_amdgcn_raw_buffer_load_b128(seq_rsrc, i * 4, 0, 0);
  } else if (BASE == B_RLE) {
    // rows i..i+3 share a bitmap word (i is a multiple of 4): run index of row i from the rank
    // table, the next three advance by their start flags (templated/RLE.hpp:164-191)
    const u32 w = i >> 6, p = i & 63;
    const u64 word = lds64(e.rk.bits_off + 8 * w);
    const u32 r = lds32(e.rk.rank_off + 4 * w) + __popcll(word & ((2ull << p) - 1));
    const u32 m = (u32)(word >> (p + 1));
    const u32 last = e.tab.count - 1;
    u32 r0 = r ? r - 1 : 0;
    u32 r1 = r0 + (m & 1), r2 = r1 + ((m >> 1) & 1), r3 = r2 + ((m >> 2) & 1);
    r0 = r0 < last ? r0 : last;
    r1 = r1 < last ? r1 : last;
    r2 = r2 < last ? r2 : last;
    r3 = r3 < last ? r3 : last;
    if (TAB_LDS) {
      v.x = lds32(e.tab.lds + 4 * r0);
      v.y = lds32(e.tab.lds + 4 * r1);
      v.z = lds32(e.tab.lds + 4 * r2);
      v.w = lds32(e.tab.lds + 4 * r3);
    } else {
      v.x = tab_ld32(e.tab, r0);
      v.y = tab_ld32(e.tab, r1);
      v.z = tab_ld32(e.tab, r2);
      v.w = tab_ld32(e.tab, r3);
    }
  } else {  // B_FREQ: top value unless the row is a bitmap member (templated/Frequency.hpp:118-135)
    const u32 w = i >> 6, p = i & 63;
    const u64 word = lds64(e.rk.bits_off + 8 * w);
    const u32 m = (u32)(word >> p) & 15;
    v.x = v.y = v.z = v.w = e.k0;
    if (m) {
      const u32 last = e.tab.count - 1;
      u32 r = lds32(e.rk.rank_off + 4 * w) + __popcll(word & ((1ull << p) - 1));
      auto ld = [&](u32 q) { return TAB_LDS ? lds32(e.tab.lds + 4 * (q < last ? q : last)) : tab_ld32(e.tab, q < last ? q : last); };
      if (m & 1) v.x = ld(r);
      r += m & 1;
      if (m & 2) v.y = ld(r);
      r += (m >> 1) & 1;
      if (m & 4) v.z = ld(r);
      r += (m >> 2) & 1;
      if (m & 8) v.w = ld(r);
    }
  }
  return v;
}

constexpr u32 kUnroll4 = 2;
// The decoded column is stored non-temporally. With ordinary write-back stores a launch that reads
// little (a fill, run expansion) ends up to 18 % sooner, but its dirty lines are written back during
// the launch behind it, which loses the same time: over a step the sum is the same, and with several
// launches beside each other (a relation-level plan) write-back stores were 2-3 % slower.
#ifdef BTR_PLAIN_STORES  // experiment builds only
constexpr int kStoreNt = 0;
#define BTR_NT_STORE(v, p) (*(p) = (v))
#else
constexpr int kStoreNt = 2;  // cache policy bit of buffer stores: non-temporal
#define BTR_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif

template <u32 BASE, u32 PS, bool TAB_LDS = false>
__device__ __forceinline__ void mat_int4(const Ctx& c, const IntEval& e, u32* __restrict__ dst, u32 count) {
  const u32 tid = threadIdx.x;
  const u32 quads = count & ~3u;
  const __amdgpu_buffer_rsrc_t seq_rsrc =
      BASE == B_SEQ ? (e.tab.loc == LOC_LDS ? make_rsrc(nullptr, 0) : make_rsrc(e.tab.g, e.tab.count * 4))
                    : make_rsrc(nullptr, 0);
  u32 bad = 0;
  // a plain copy has nothing but its loads to wait for: more of them in flight
  constexpr u32 kUnrollHere = BASE == B_SEQ ? 2 * kUnroll4 : kUnroll4;
  constexpr u32 kRowsPerIter = kBlockThreads * 4 * kUnrollHere;
  // whole iterations need no row clamps and no store predicates
  const u32 whole = quads / kRowsPerIter * kRowsPerIter;
  auto iteration = [&](u32 base, auto tag) {
    constexpr bool PARTIAL = decltype(tag)::value;
    u32x4 v[kUnrollHere];
#pragma unroll
    for (u32 u = 0; u < kUnrollHere; u++) {
      const u32 i = base + (u * kBlockThreads + tid) * 4;
      v[u] = eval_int4_base<BASE, TAB_LDS>(e, PARTIAL ? (i < quads ? i : quads - 4) : i, seq_rsrc);
    }
#pragma unroll
    for (u32 u = 0; u < kUnrollHere; u++) {
      v[u].x = apply_posts<PS>(e, v[u].x, bad);
      v[u].y = apply_posts<PS>(e, v[u].y, bad);
      v[u].z = apply_posts<PS>(e, v[u].z, bad);
      v[u].w = apply_posts<PS>(e, v[u].w, bad);
    }
#pragma unroll
    for (u32 u = 0; u < kUnrollHere; u++) {
      const u32 i = base + (u * kBlockThreads + tid) * 4;
      // non-temporal: the decoded column is written once and not read back by this kernel; keeping
      // it out of the caches also keeps its write-back out of the next launch's time
      if (!PARTIAL || i < quads) BTR_NT_STORE(v[u], reinterpret_cast<u32x4*>(dst + i));
    }
  };
  for (u32 base = 0; base < whole; base += kRowsPerIter) iteration(base, std::false_type{});
  if (whole < quads) iteration(whole, std::true_type{});
  if (tid < count - quads) dst[quads + tid] = eval_int<BASE, false, PS>(e, quads + tid, bad);
  report(c, bad);
}

// Whether a root pipe can take the four-rows-per-lane path into a 16-byte aligned output.
__device__ __forceinline__ bool wide_rows(const IntEval& e, const Tab& out, u32 count) {
  if (out.loc == LOC_LDS || (reinterpret_cast<u64>(out.g) & 15) != 0 || count < 4) return false;
  if (e.base == B_SEQ) return e.tab.loc != LOC_LDS && e.tab.esize == 4;
  if (e.base == B_RLE || e.base == B_FREQ) return e.tab.count > 0 || e.base == B_FREQ;
  return e.base == B_ONE;
}

template <u32 BASE>
__device__ __forceinline__ void mat_int4_ps(const Ctx& c, const IntEval& e, const Tab& out, u32 count) {
  u32* dst = reinterpret_cast<u32*>(const_cast<u8*>(out.g));
  if ((BASE == B_RLE || BASE == B_FREQ) && e.tab.loc == LOC_LDS && e.tab.count > 0) {
    switch (post_signature(e)) {
      case PS_NONE: mat_int4<BASE, PS_NONE, true>(c, e, dst, count); return;
      case PS_ADD: mat_int4<BASE, PS_ADD, true>(c, e, dst, count); return;
      case PS_DICT_LDS: mat_int4<BASE, PS_DICT_LDS, true>(c, e, dst, count); return;
      default: break;
    }
  }
  switch (post_signature(e)) {
    case PS_NONE: mat_int4<BASE, PS_NONE>(c, e, dst, count); break;
    case PS_ADD: mat_int4<BASE, PS_ADD>(c, e, dst, count); break;
    case PS_DICT_LDS: mat_int4<BASE, PS_DICT_LDS>(c, e, dst, count); break;
    default: mat_int4<BASE, PS_GEN>(c, e, dst, count); break;
  }
}

// Widths and first words of the four miniblocks of a group, per lane: lanes 0-31 work on
// miniblocks 0 and 2, lanes 32-63 on miniblocks 1 and 3 (values lane and lane + 64 of the group).
struct LaneGroup {
  u32 w_lo, b_lo, w_hi, b_hi;
};
__device__ __forceinline__ LaneGroup lane_group(u64 index_entry, u32 upper_mask) {
  const u32 off = uniform((u32)index_entry), hdr = uniform((u32)(index_entry >> 32));
  u32 b0 = hdr >> 24, b1 = (hdr >> 16) & 0xFF, b2 = (hdr >> 8) & 0xFF, b3 = hdr & 0xFF;
  b0 = b0 > 32 ? 32 : b0;
  b1 = b1 > 32 ? 32 : b1;
  b2 = b2 > 32 ? 32 : b2;
  b3 = b3 > 32 ? 32 : b3;
  const u32 first = off + 1;
  LaneGroup g;  // branch-free selects: upper_mask is all ones in lanes 32-63
  g.w_lo = first + (b0 & upper_mask);
  g.b_lo = (b0 & ~upper_mask) | (b1 & upper_mask);
  g.w_hi = first + b0 + b1 + (b2 & upper_mask);
  g.b_hi = (b2 & ~upper_mask) | (b3 & upper_mask);
  return g;
}

// Groups [g_lo, g_hi) of a bit-packed root read from global memory: every wave takes whole
// 128-value groups.
constexpr u32 kGroupsPerIter = 4;

template <u32 PS>
__device__ __forceinline__ void eval_bp_groups(const Ctx& c, const IntEval& e, const Tab& out, u32 g_lo,
                                               u32 g_hi) {
  const u32 lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const BpView& bp = e.bp;
  const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(bp.base, bp.bytes);
  const u32 k = lane & 31;
  const u32 upper_mask = lane >= 32 ? ~0u : 0u;
  u32 bad = 0;
  for (u32 g0 = g_lo + wave * kGroupsPerIter; g0 < g_hi; g0 += (kBlockThreads / 64) * kGroupsPerIter) {
    u32 v[kGroupsPerIter][2];
#pragma unroll
    for (u32 j = 0; j < kGroupsPerIter; j++) {
      const u32 g = g0 + j < g_hi ? g0 + j : g_hi - 1;
      const LaneGroup lg = lane_group(lds64(bp.index_off + 8 * g), upper_mask);
      v[j][0] = bp_extract(rsrc, lg.w_lo, lg.b_lo, k);
      v[j][1] = bp_extract(rsrc, lg.w_hi, lg.b_hi, k);
    }
#pragma unroll
    for (u32 j = 0; j < kGroupsPerIter; j++) {
      v[j][0] = apply_posts<PS>(e, v[j][0], bad);
      v[j][1] = apply_posts<PS>(e, v[j][1], bad);
    }
#pragma unroll
    for (u32 j = 0; j < kGroupsPerIter; j++) {
      const u32 i = (g0 + j) * btrfmt::FBP_GROUP + lane;
      if (g0 + j < g_hi) {
        if (i < bp.rounded) tab_st32(out, i, v[j][0]);
        if (i + 64 < bp.rounded) tab_st32(out, i + 64, v[j][1]);
      }
    }
  }
  report(c, bad);
}

template <u32 PS>
__device__ __forceinline__ void eval_bp_tail(const Ctx& c, const IntEval& e, const Tab& out, u32 count) {
  const BpView& bp = e.bp;
  u32 bad = 0;
  for (u32 i = bp.rounded + threadIdx.x; i < count; i += kBlockThreads)
    tab_st32(out, i, apply_posts<PS>(e, lds32(bp.tail_off + 4 * ((i - bp.rounded) & 31)), bad));
  report(c, bad);
}

template <u32 PS>
__device__ __forceinline__ void mat_int_bp(const Ctx& c, const IntEval& e, const Tab& out, u32 count) {
  eval_bp_groups<PS>(c, e, out, 0, (e.bp.rounded + btrfmt::FBP_GROUP - 1) / btrfmt::FBP_GROUP);
  eval_bp_tail<PS>(c, e, out, count);
}

// ---- root step over a bit-packed stream, payload staged through an LDS ring ------------------------
// Used by the stream kernels (512 threads). The compressed stream is read from HBM exactly once,
// by 16-byte-per-lane direct-to-LDS buffer loads (1 KiB per wave instruction); everything else
// works out of LDS, with the waves of the workgroup in fixed roles:
//   last wave (loader)  keeps kRingAhead batches of 1024 payload words in flight into a ring of
//                       kRingBatches batches;
//   wave 0 (walker)     follows the group-header chain through batch t+1 and writes {word offset,
//                       header} per group into the index, while
//   the other waves     unpack the groups that start in batch t from the ring, apply the post-op
//                       and store the rows.
// One workgroup barrier per batch: "batch t+1 has landed and the groups of batch t are indexed"
// (a group that starts in batch t may run into batch t+1).
constexpr u32 kStreamThreads = 512;
constexpr u32 kStreamWaves = kStreamThreads / 64;
constexpr u32 kUnpackWaves = kStreamWaves - 2;
// The double-dictionary kernel keeps up to 32 KiB of dictionary in LDS, so few of its workgroups
// fit a CU: it runs 16 waves per workgroup (14 unpackers) to fill the CU with two of them.
constexpr u32 kDictdThreads = 1024;
constexpr u32 kDictdWaves = kDictdThreads / 64;
constexpr u32 kRingBatches = 4;  // power of two
constexpr u32 kRingAhead = 3;    // batches issued ahead of the one being unpacked
constexpr u32 kRingWords = kRingBatches * kBatchWords;
static_assert(kRingWords * 4 == kStreamRingBytes, "ring size is shared with the plan compiler");

__device__ __forceinline__ void barrier_lds() {  // LDS accesses of this wave complete, then barrier
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// A ring starts at an LDS offset that is a multiple of its size (RING); the word behind it mirrors
// ring word 0, so that the two consecutive words an extraction needs can always be read at a and
// a + 4.
template <u32 RING = 0>
__device__ __forceinline__ u32 ring_word(u32 w) {
  return lds32(RING | ((w << 2) & (kStreamRingBytes - 1)));
}

// Two consecutive words of ring 0 (which starts at LDS address 0) at byte offset a.
__device__ __forceinline__ u32x2 ring_pair(u32 a) {
  typedef __attribute__((address_space(3))) const u32 lds_word;
  lds_word* p = (lds_word*)(__UINTPTR_TYPE__)a;
  u32x2 r;
  r.x = p[0], r.y = p[1];
  return r;
}

// Batch t of a payload -> its ring slot, by direct-to-LDS loads (1 KiB per wave instruction).
template <u32 RING>
__device__ __forceinline__ void ring_issue(__amdgpu_buffer_rsrc_t rsrc, u32 t, u32 batches) {
  if (t >= batches) return;
  const u32 lane = threadIdx.x & 63;
  const u32 slot = RING + (t & (kRingBatches - 1)) * (kBatchWords * 4);
#pragma unroll
  for (u32 k = 0; k < kWalkBatch; k++)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds_raw + slot + k * 1024),
                                             16, (t * kBatchWords + 256 * k + 4 * lane) * 4, 0, 0, 0);
}
// Refreshes the mirror word; called by the loading wave after a batch has landed in slot 0.
template <u32 RING>
__device__ __forceinline__ void ring_mirror() {
  if ((threadIdx.x & 63) == 0) lds_st32(RING + kStreamRingBytes, lds32(RING));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// The loading wave's wait: everything but the youngest `keep_batches` batches has landed.
__device__ __forceinline__ void ring_wait(bool keep_one_batch) {
  if (keep_one_batch)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// Lane-constant selectors of the unpackers: lane l produces values l and l + 64 of a group, i.e.
// position l % 32 of miniblocks l / 32 and l / 32 + 2.
struct UnpackLane {
  u32 k;             // position inside the miniblock
  u32 sh_lo, sh_hi;  // bit position of the miniblock's width byte in the group header
  u32 pm_lo, pm_hi;  // header bytes of the miniblocks in front of it
  u32 row4;          // byte offset of the lane's first row inside the group's 512 output bytes
};
__device__ __forceinline__ UnpackLane unpack_lane() {
  const u32 lane = threadIdx.x & 63, h = lane >> 5;
  UnpackLane u;
  u.k = lane & 31;
  u.sh_lo = 24 - 8 * h;
  u.sh_hi = 8 - 8 * h;
  u.pm_lo = h ? 0xFF000000u : 0u;
  u.pm_hi = h ? 0xFFFFFF00u : 0xFFFF0000u;
  u.row4 = lane * 4;
  return u;
}

// One value out of the ring. `first` is the first payload word of the group, `hdr` its header
// (b0<<24)|(b1<<16)|(b2<<8)|b3; the miniblock is picked by (sh, pm). Value k of a miniblock of
// width b occupies bits [k*b, (k+1)*b) of the miniblock's little-endian bit stream. A width above
// 32 (flagged by the walker) yields garbage but stays inside the ring.
template <u32 RING = 0>
__device__ __forceinline__ u32 ring_extract(u32 first, u32 hdr, u32 sh, u32 pm, u32 k) {
  const u32 b = __builtin_amdgcn_ubfe(hdr, sh, 8);
  const u32 w = __builtin_amdgcn_sad_u8(hdr & pm, 0u, first);
  const u32 bit = __umul24(k, b);
  const u32 a = RING | (((w + (bit >> 5)) << 2) & (kStreamRingBytes - 1));
  const u32 lo = lds32(a), hi = lds32(a + 4);
  const u32 x = __builtin_amdgcn_alignbit(hi, lo, bit);
  const u32 full = (u32)__builtin_amdgcn_sbfe(b, 5, 1);  // all ones when b == 32
  return __builtin_amdgcn_ubfe(x, 0u, b) | (x & full);
}

// Post-op of a streamed root, resolved once per workgroup.
struct StreamPost {
  u32 bias;                       // PS_ADD
  u32 n, dict_off;                // PS_DICT_*: dictionary size; LDS offset of its copy
  __amdgpu_buffer_rsrc_t dict_g;  // PS_DICT_GLB: the dictionary inside the compressed bytes
};
// Dictionary codes are not clamped per value: the largest one is tracked in `code_max` and
// checked once; an LDS read is clamped by one min, a buffer load past the end returns 0.
template <u32 PS>
__device__ __forceinline__ u32 stream_post(const StreamPost& sp, u32 v, u32& code_max) {
  if (PS == PS_ADD) return v + sp.bias;
  if (PS == PS_DICT_LDS) {
    code_max = code_max > v ? code_max : v;
    v = v < sp.n ? v : sp.n - 1;
    return lds32(sp.dict_off + 4 * v);
  }
  if (PS == PS_DICT_GLB) {
    code_max = code_max > v ? code_max : v;
    return __builtin_amdgcn_raw_buffer_load_b32(sp.dict_g, v << 2, 0, 0);
  }
  return v;
}

// Groups g_lo + uw, + kUnpackWaves, ... below g_hi, two per iteration. Rows go out through a
// buffer resource that ends at the last packed row, so a partial last group needs no masking.
template <u32 PS>
__device__ __forceinline__ void ring_eval_groups(const StreamPost& sp, const UnpackLane& u,
                                                 __amdgpu_buffer_rsrc_t dst_rsrc, u32 index_off, u32 g_lo, u32 g_hi,
                                                 u32 uw, u32& code_max) {
  const u32 lane = threadIdx.x & 63;
  for (u32 g = g_lo + uw; g < g_hi; g += 2 * kUnpackWaves) {
    const u32 g2 = g + kUnpackWaves;
    const bool two = g2 < g_hi;
    const u64 e1 = lds64(index_off + 8 * g), e2 = lds64(index_off + 8 * (two ? g2 : g));
    // the index entries are wave-uniform: when the four miniblocks of a group share one width b < 32
    // its 128 values are one bit string, value v at bit v * b, and the lane's two values (v = lane
    // and lane + 64) sit 2 * b words apart at the same bit offset
    const u32 f1 = uniform((u32)e1), h1 = uniform((u32)(e1 >> 32)), f2 = uniform((u32)e2), h2 = uniform((u32)(e2 >> 32));
    const u32 b1 = h1 & 0xFF, b2 = h2 & 0xFF;
    u32 v[4];
    if (h1 == b1 * 0x01010101u && h2 == b2 * 0x01010101u && b1 < 32 && b2 < 32) {
#pragma unroll
      for (u32 j = 0; j < 2; j++) {
        const u32 b = j ? b2 : b1, f = j ? f2 : f1;
        u32 bit;
        asm("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(bit) : "s"(b), "v"(lane));
        const u32 a = (f + (bit >> 5)) << 2;
        const u32x2 lo = ring_pair(a & (kStreamRingBytes - 1)), hi = ring_pair((a + 8 * b) & (kStreamRingBytes - 1));
        v[2 * j] = __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(lo.y, lo.x, bit), 0u, b);
        v[2 * j + 1] = __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(hi.y, hi.x, bit), 0u, b);
      }
    } else {
      v[0] = ring_extract(f1, h1, u.sh_lo, u.pm_lo, u.k);
      v[1] = ring_extract(f1, h1, u.sh_hi, u.pm_hi, u.k);
      v[2] = ring_extract(f2, h2, u.sh_lo, u.pm_lo, u.k);
      v[3] = ring_extract(f2, h2, u.sh_hi, u.pm_hi, u.k);
    }
#pragma unroll
    for (u32 j = 0; j < 4; j++) v[j] = stream_post<PS>(sp, v[j], code_max);
    const u32 o1 = g * (btrfmt::FBP_GROUP * 4) + u.row4;
    const u32 o2 = (two ? g2 * (btrfmt::FBP_GROUP * 4) : 0x80000000u) + u.row4;  // out of range: dropped
    __builtin_amdgcn_raw_buffer_store_b32(v[0], dst_rsrc, o1, 0, kStoreNt);
    __builtin_amdgcn_raw_buffer_store_b32(v[1], dst_rsrc, o1 + 256, 0, kStoreNt);
    __builtin_amdgcn_raw_buffer_store_b32(v[2], dst_rsrc, o2, 0, kStoreNt);
    __builtin_amdgcn_raw_buffer_store_b32(v[3], dst_rsrc, o2 + 256, 0, kStoreNt);
  }
}

// One batch of the header chain, read from the ring. Index entries are {first payload word, header}.
// While groups are equally long the chain is followed by run steps on the ring itself: lane j reads
// the word at pos + j*D (D = length of the group at pos) and the leading lanes that find the same
// length there hold real headers. Only when a run breaks inside the batch are the lengths of all
// 1024 words computed for the pointer-doubling walk.
template <u32 RING = 0>
__device__ __forceinline__ void ring_walk_batch(u32 index_off, u32 jump_off, u32 t, u32 groups, ChainState& st,
                                                u32& bad_hdr) {
  const u32 lane = threadIdx.x & 63;
  const u32 bb = t * kBatchWords;
  const u32 slot = RING + (t & (kRingBatches - 1)) * (kBatchWords * 4);
  auto glen = [](u32 x) {
    const u32 s = 1 + __builtin_amdgcn_sad_u8(x, 0u, 0u);
    return s > 255 ? 255u : s;
  };
  auto put = [&](u32 g, u32 hv, u32 position) {
    if ((hv & 0xC0C0C0C0u) || ((hv + 0x5F5F5F5Fu) & 0x80808080u)) bad_hdr = 1;  // widths must be <= 32
    lds_st64(index_off + 8 * g, ((u64)hv << 32) | (position + 1));
  };
  while (st.spec_ok && st.pos < bb + kBatchWords && st.g < groups) {
    const u32 D = uniform(glen(ring_word<RING>(st.pos)));
    const u32 pj = st.pos + lane * D;
    const bool inside = pj < bb + kBatchWords;
    const u32 hv = ring_word<RING>(inside ? pj : st.pos);
    const u64 ok = __ballot(inside && glen(hv) == D);
    u32 m = (u32)__builtin_ctzll(~ok | (1ull << 63));
    m = m == 63 && (ok >> 63) ? 64 : m;
    m = m < groups - st.g ? m : groups - st.g;
    if (lane < m) put(st.g + lane, hv, pj);
    st.g = uniform(st.g + m);
    st.pos = uniform(st.pos + m * D);
    if (m < 64 && st.pos < bb + kBatchWords) st.spec_ok = 0;  // the run broke inside the batch
  }
  if (!(st.pos < bb + kBatchWords && st.g < groups)) return;
  u32 d[kWalkBatch];
#pragma unroll
  for (u32 k = 0; k < kWalkBatch; k++)
    d[k] = group_lengths(*reinterpret_cast<const u32x4*>(lds_raw + slot + k * 1024 + lane * 16));
  walk_chain(d, bb, groups, jump_off, st, [&](u32 first, u32 count, u32 position) {
    if (lane < count) put(first + lane, ring_word<RING>(position), position);
  });
}

// One pass of a bit-packed payload through the ring, by all waves of a 512-thread workgroup.
// After walking a batch the walker stores publish(chain state) in LDS; in iteration t the unpack
// waves get unpack(previous value, value after batch t, unpack wave number). With publish = groups
// found, [lo, hi) are the groups that start in batch t: their words are in the ring (a group that
// starts in batch t may run into batch t+1, which has landed) and their index entries are written.
// Ends with the walker's tail decode and a workgroup barrier.
#ifdef BTR_DEBUG_STEPS
// Diagnostic builds: time (100 MHz ticks) a wave spent in a section, summed over the batches of a
// pass and stored as "item start + sum" so that the stamp report prints the sum.
#define BTR_ACC_BEGIN() const u32 acc_t0_ = (u32)wall_clock64()
#define BTR_ACC_END(acc) (acc) += (u32)wall_clock64() - acc_t0_
#define BTR_ACC_STORE(c, slot, acc)                                                                       \
  do {                                                                                                    \
    if (((c).debug & 8) && (slot) && (threadIdx.x & 63) == 0 && (slot) < kDebugSlots)                     \
      reinterpret_cast<u32*>((c).debug_buf)[slot] = reinterpret_cast<u32*>((c).debug_buf)[0] + (acc);     \
  } while (0)
#else
#define BTR_ACC_BEGIN() do {} while (0)
#define BTR_ACC_END(acc) do {} while (0)
#define BTR_ACC_STORE(c, slot, acc) do {} while (0)
#endif

// EARLY_RELEASE (with KEEP_PREVIOUS): only the first unpack unit of iteration t may still read batch
// t-1 (units lag the digits by less than one unit). The wave that takes that unit stores t + 1 into
// the release word (ring_release_word) once its ring reads are done, and the loader refills the
// slot of batch t-1 right then instead of at the next barrier: two batches in flight instead of one.
// publish() values must then be unit counts boundaries such that "lo == hi" means "no unit".
__device__ __forceinline__ u32 ring_release_word(const Ctx& c) { return c.red_off + 40; }

// The batches a pass starts with (as many as it keeps in flight), issued by its loading wave. A caller
// may do this itself ahead of the pass (PREISSUED), behind a barrier after which nothing reads the ring.
template <bool KEEP_PREVIOUS, bool EARLY_RELEASE>
__device__ __forceinline__ void ring_issue_first(const BpView& bp, const Pipe& p) {
  constexpr u32 ahead = KEEP_PREVIOUS && !EARLY_RELEASE ? kRingAhead - 1 : kRingAhead;
  const u32 batches = (uniform(p.a[1]) + kBatchWords - 1) / kBatchWords;
  const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(bp.base, bp.bytes);
  for (u32 t = 0; t < ahead; t++) ring_issue<0>(rsrc, t, batches);
}

template <bool KEEP_PREVIOUS = false, u32 WAVES = kStreamWaves, u32 DBG = 0, bool EARLY_RELEASE = false,
          bool PREISSUED = false, typename Publish, typename Unpack>
__device__ __forceinline__ void ring_pass(const Ctx& c, const Pipe& p, const BpView& bp, u32 jump_off, Publish publish,
                                          Unpack unpack) {
  static_assert(!EARLY_RELEASE || KEEP_PREVIOUS, "early release refines the keep-previous schedule");
  u32 acc = 0;
  (void)acc;
  const u32 lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const u32 W = uniform(p.a[1]), groups = uniform(p.a[5]);
  const u32 batches = (W + kBatchWords - 1) / kBatchWords;
  const u32 slot_off = c.red_off + 32;  // two LDS words: published value after batch t, slot t & 1
  // KEEP_PREVIOUS: batch t-1 stays readable during iteration t (for unpackers that lag behind
  // the walker), at the price of one batch less in flight.
  constexpr u32 ahead = KEEP_PREVIOUS && !EARLY_RELEASE ? kRingAhead - 1 : kRingAhead;
  constexpr bool kTwoInFlight = !KEEP_PREVIOUS || EARLY_RELEASE;

  if (wave == WAVES - 1) {
    // ---- loader
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(bp.base, bp.bytes);
    if (!PREISSUED)
      for (u32 t = 0; t < ahead; t++) ring_issue<0>(rsrc, t, batches);
    if (EARLY_RELEASE && lane == 0) lds_st32(ring_release_word(c), 0);
    // batches 0 and 1 landed (only batch 2 may still be in flight), then the start barrier
    ring_wait(kTwoInFlight && batches > 2);
    ring_mirror<0>();
    __builtin_amdgcn_s_barrier();
    u32 prev_published = 0;
    for (u32 t = 0; t < batches; t++) {
      // before barrier t: batch t+1 landed; at most batch t+2 stays in flight
      {
        BTR_ACC_BEGIN();
        ring_wait(kTwoInFlight && t + 2 < batches);
        BTR_ACC_END(acc);
      }
      if (((t + 1) & (kRingBatches - 1)) == 0) ring_mirror<0>();
      barrier_lds();
      if (EARLY_RELEASE) {
        // the slot of batch t + 3 holds batch t - 1: wait until this iteration's first unit (if there
        // is one) has taken its words out of it
        const u32 published = uniform(lds32(slot_off + 4 * (t & 1)));
        if (t + ahead < batches && published != prev_published) {
          // bounded: a release that never comes must not hang the GPU; the batch is then issued over
          // words that may still be read, so the chunk is reported as not decoded
          bool released = false;
          for (u32 spin = 0; spin < (1u << 20); spin++) {
            if (uniform(*reinterpret_cast<volatile const u32*>(lds_raw + ring_release_word(c))) == t + 1) {
              released = true;
              break;
            }
            __builtin_amdgcn_s_sleep(2);
          }
          if (!released && lane == 0) raise(c, ST_INTERNAL);
        }
        prev_published = published;
      }
      ring_issue<0>(rsrc, t + ahead, batches);  // its slot held batch t + ahead - 4, no longer read
    }
    BTR_ACC_STORE(c, DBG == 9 ? 8 : DBG, acc);  // loader: waiting for its loads
  } else if (wave == 0) {
    // ---- walker: batch 0 before barrier 0, then batch t+1 during iteration t
    ChainState st{1, 0, 1};
    u32 bad_hdr = 0;
    const u32 tail_word = load_tail_words(make_rsrc(bp.base, bp.bytes), W);
    __builtin_amdgcn_s_setprio(3);  // the chain is serial: keep it ahead of the unpackers
    jump_tables_init(jump_off);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ring_walk_batch(bp.index_off, jump_off, 0, groups, st, bad_hdr);
    {
      const u32 v = publish(st);
      if (lane == 0) lds_st32(slot_off, v);
    }
    for (u32 t = 0; t < batches; t++) {
      barrier_lds();
      if (t + 1 < batches) {
        BTR_ACC_BEGIN();
        ring_walk_batch(bp.index_off, jump_off, t + 1, groups, st, bad_hdr);
        const u32 v = publish(st);
        if (lane == 0) lds_st32(slot_off + 4 * ((t + 1) & 1), v);
        BTR_ACC_END(acc);
      }
    }
    BTR_ACC_STORE(c, DBG == 9 ? 15 : DBG ? DBG + 1 : 0, acc);  // walker: walking and publishing
    base_priority(c);
    // sentinel, flags and the VariableByte tail
    walker_finish(WalkEnd{W, bp.rounded, groups, bp.index_off, st.pos, st.g, __any(bad_hdr) ? 1u : 0u, tail_word}, c, p);
  } else {
    // ---- unpackers
    u32 lo = 0;
    __builtin_amdgcn_s_barrier();
    for (u32 t = 0; t < batches; t++) {
      barrier_lds();
      const u32 hi = uniform(lds32(slot_off + 4 * (t & 1)));
      BTR_ACC_BEGIN();
      unpack(lo, hi, wave - 1);
      BTR_ACC_END(acc);
      lo = hi;
    }
    if (DBG == 9)
      BTR_ACC_STORE(c, 8 + wave, acc);  // every unpacker of the row pass: slots 9..14
    else if (wave == 1)
      BTR_ACC_STORE(c, DBG ? DBG + 2 : 0, acc);  // first unpacker: unpacking
  }
  __syncthreads();
}

template <u32 PS>
__device__ __forceinline__ void stream_int_bp(const Ctx& c, const Pipe& p, const IntEval& e, u32* __restrict__ dst,
                                              u32 jump_off) {
  const UnpackLane u = unpack_lane();
  const __amdgpu_buffer_rsrc_t dst_rsrc = make_rsrc(dst, e.bp.rounded * 4);
  StreamPost sp;
  sp.bias = e.post_bias[0];
  sp.n = e.post_tab[0].count;
  sp.dict_off = e.post_tab[0].lds;
  sp.dict_g = make_rsrc(PS == PS_DICT_GLB ? e.post_tab[0].g : e.bp.base, PS == PS_DICT_GLB ? sp.n * 4 : 0);
  u32 code_max = 0;
  ring_pass<false, kStreamWaves, 0, false, true>(  // first batches issued by the caller
      c, p, e.bp, jump_off, [](const ChainState& st) { return st.g; },
      [&](u32 g_lo, u32 g_hi, u32 uw) { ring_eval_groups<PS>(sp, u, dst_rsrc, e.bp.index_off, g_lo, g_hi, uw, code_max); });
  if ((PS == PS_DICT_LDS || PS == PS_DICT_GLB) && code_max >= sp.n) raise(c, ST_DICT_CODE);
  {
    u32 bad = 0;
    for (u32 i = e.bp.rounded + threadIdx.x; i < e.count; i += kStreamThreads)
      dst[i] = apply_posts<PS>(e, lds32(e.bp.tail_off + 4 * ((i - e.bp.rounded) & 31)), bad);
    report(c, bad);
  }
}

// ---- DICT(BP) doubles (templated/DynamicDictionary.hpp:162-216): codes through the ring, 8-byte gather --
__device__ __forceinline__ void stream_dictd(const Ctx& c, const Pipe& root, u64* __restrict__ dst, u32 jump_off) {
  const Pipe& pc = pipe_at(c, uniform(root.a[0]));
  const BpView bp = resolve_bp(c, pc);
  const Tab dict = resolve_tab(c, uniform(root.a[1]));
  const u32 n = uniform(root.count);
  const UnpackLane u = unpack_lane();
  const __amdgpu_buffer_rsrc_t dst_rsrc = make_rsrc(dst, bp.rounded * 8);
  const bool in_lds = dict.loc == LOC_LDS;
  u32 code_max = 0;
  auto gather = [&](u32 code) {
    code_max = code_max > code ? code_max : code;
    code = code < dict.count ? code : dict.count - 1;
    return in_lds ? lds64(dict.lds + 8 * code) : ldg64(dict.g + 8ull * code);
  };
  ring_pass<false, kDictdWaves, 0, false, true>(  // first batches issued by the caller
      c, pc, bp, jump_off, [](const ChainState& st) { return st.g; },
      [&](u32 g_lo, u32 g_hi, u32 uw) {
        for (u32 g = g_lo + uw; g < g_hi; g += kDictdWaves - 2) {
          const u64 en = lds64(bp.index_off + 8 * g);
          const u32 c0 = ring_extract<0>((u32)en, (u32)(en >> 32), u.sh_lo, u.pm_lo, u.k);
          const u32 c1 = ring_extract<0>((u32)en, (u32)(en >> 32), u.sh_hi, u.pm_hi, u.k);
          const u64 r0 = gather(c0), r1 = gather(c1);
          const u32 o = g * (btrfmt::FBP_GROUP * 8) + 2 * u.row4;
          u32x2 s0, s1;
          s0.x = (u32)r0, s0.y = (u32)(r0 >> 32), s1.x = (u32)r1, s1.y = (u32)(r1 >> 32);
          __builtin_amdgcn_raw_buffer_store_b64(s0, dst_rsrc, o, 0, kStoreNt);
          __builtin_amdgcn_raw_buffer_store_b64(s1, dst_rsrc, o + 512, 0, kStoreNt);
        }
      });
  for (u32 i = bp.rounded + threadIdx.x; i < n; i += kDictdThreads)
    dst[i] = gather(lds32(bp.tail_off + 4 * ((i - bp.rounded) & 31)));
  if (code_max >= dict.count) raise(c, ST_DICT_CODE);
}

// ---- PSEUDODECIMAL(BP, BP, raw patches): digits streamed through the ring -------------------------
// double/Pseudodecimal.cpp:424-436: row i is patches[x] when its exponent is the exception code
// (x = exceptions before row i), else double(int32 digits[i - x]) * 10^-exponent. The digit
// stream is the wide one and goes through the LDS ring exactly like a bit-packed integer root;
// the exponents (a few bits per row) were indexed and scanned for exception codes by the
// preparation steps and are bit-unpacked from global memory, where they stay cache resident.
// Rows are produced in 128-row groups of the exponent stream; the digit of a row is gathered per
// lane from wherever it sits in the ring. After each batch the walker converts "digits indexed
// so far" into "rows that can be produced" by a select on the exception bitmap.
__device__ __forceinline__ u64 pseudo_value(u32 digits, u32 ex, u32 frac_off, u32& exbad) {
  u32 idx = ex & btrfmt::DEC_INDEX_MASK;
  exbad |= (ex != btrfmt::DEC_EXCEPTION_CODE && idx > btrfmt::DEC_MAX_EXPONENT) ? 2u : 0u;
  idx = idx > btrfmt::DEC_MAX_EXPONENT ? btrfmt::DEC_MAX_EXPONENT : idx;
  const double f = __longlong_as_double((long long)lds64(frac_off + 8 * idx));
  return (u64)__double_as_longlong((double)(int)digits * f);
}

// Same extraction as ring_extract, from global memory (generic index entry: header position).
__device__ __forceinline__ u32 global_extract(__amdgpu_buffer_rsrc_t rsrc, u32 first, u32 hdr, u32 sh, u32 pm, u32 k) {
  const u32 b = __builtin_amdgcn_ubfe(hdr, sh, 8);
  const u32 w = __builtin_amdgcn_sad_u8(hdr & pm, 0u, first);
  const u32 bit = __umul24(k, b);
  const u32x2 ww = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (w + (bit >> 5)) * 4, 0, 0);
  const u32 x = __builtin_amdgcn_alignbit(ww.y, ww.x, bit);
  const u32 full = (u32)__builtin_amdgcn_sbfe(b, 5, 1);
  return __builtin_amdgcn_ubfe(x, 0u, b) | (x & full);
}

// First row that needs digit number `target` or a later one: the position of the target-th
// clear bit of the exception bitmap (rows before it only use digits below target), or `n` when
// there are not that many. One wave; words = n / 64 + 1 <= 4096.
__device__ __forceinline__ u32 select_clear(const RankView& rk, u32 words, u32 n, u32 target) {
  const u32 lane = threadIdx.x & 63;
  const u32 stride = (words + 63) / 64;
  auto clear_before = [&](u32 w) { return 64 * w - lds32(rk.rank_off + 4 * w); };
  // largest word w with clear_before(w) <= target, in two 64-way steps
  const u32 w1 = lane * stride;
  const u32 c1 = (u32)__popcll(__ballot(w1 < words && clear_before(w1 < words ? w1 : 0) <= target));
  const u32 ws = (c1 ? c1 - 1 : 0) * stride;
  const u32 w2 = ws + lane;
  const u32 c2 = (u32)__popcll(__ballot(lane < stride && w2 < words && clear_before(w2 < words ? w2 : 0) <= target));
  const u32 w = uniform(ws + (c2 ? c2 - 1 : 0));
  const u32 rem = target - uniform(clear_before(w));
  const u64 inv = ~lds64(rk.bits_off + 8 * w);
  const bool hit = ((inv >> lane) & 1) && (u32)__popcll(inv & ((1ull << lane) - 1)) == rem;
  const u64 m = __ballot(hit);
  const u32 r = m ? 64 * w + (u32)__builtin_ctzll(m) : n;
  return r < n ? r : n;
}

// How the streamed exponent values are read. Plain: a value is the exponent, 23 marks a patch.
// Dictionary coded (DICT(BP) exponents, the dictionary holds at most 32 exponents, all valid -
// checked by the plan compiler): a value is a code; the powers-of-ten table is laid out by code and
// `exception` is the code of 23 (NONE when the dictionary has none).
struct ExpCode {
  u32 exception;  // stream value that marks a patch row
  u32 need_bits;  // bits needed to represent it: narrower miniblocks hold no patch row
  u32 codes;      // dictionary size, 0 for plain exponents
};

// Lane-constant selectors when lane l owns rows 4l..4l+3 of a 256-row span, i.e. of two consecutive
// 128-value groups of a bit-packed stream: group l / 32 of the span, miniblock (l / 8) % 4 of that
// group, positions 4 * (l % 8).. of that miniblock.
struct QuadLane {
  u32 grp;  // which of the span's two groups
  u32 sh;   // bit position of the miniblock's width byte in the group header
  u32 pm;   // header bytes of the miniblocks in front of it
  u32 k4;   // position of the lane's first value inside the miniblock
};
__device__ __forceinline__ QuadLane quad_lane() {
  const u32 lane = threadIdx.x & 63, m = (lane >> 3) & 3;
  QuadLane q;
  q.grp = lane >> 5;
  q.sh = 24 - 8 * m;
  q.pm = m ? ~(0xFFFFFFFFu >> (8 * m)) : 0u;
  q.k4 = 4 * (lane & 7);
  return q;
}
// Whether every width byte of a group header is at most 8: four consecutive values then fit the
// 32 bits one funnel shift delivers.
__device__ __forceinline__ bool widths_le8(u32 hdr) {
  return (hdr & 0xF0F0F0F0u) == 0 && ((hdr + 0x07070707u) & 0xF0F0F0F0u) == 0;
}
// At most 5: every value is below 32.
__device__ __forceinline__ bool widths_le5(u32 hdr) { return ((hdr | (hdr + 0x02020202u)) & 0xF8F8F8F8u) == 0; }
// Value k of a miniblock of width b (any width up to 32) out of two consecutive words.
__device__ __forceinline__ u32 take_bits(u32 lo, u32 hi, u32 bit, u32 b) {
  const u32 x = __builtin_amdgcn_alignbit(hi, lo, bit);
  return __builtin_amdgcn_ubfe(x, 0u, b) | (x & (u32)__builtin_amdgcn_sbfe(b, 5, 1));
}

// Exception flags of the exponent stream: one ring pass whose unpackers mark every row whose
// exponent is the exception code in the (cleared) bitmap. A wave takes 256 rows at a time, four
// consecutive rows per lane: exponents are at most 5 bits wide, so one 8-byte LDS read and one
// funnel shift fetch a lane's four values. Builds the stream-format group index of the exponents
// on the way.
__device__ __forceinline__ void stream_exception_bits(const Ctx& c, const Pipe& pe, const BpView& be, const RankView& rk,
                                                      u32 words, u32 n, u32 jump_off, const ExpCode& xc) {
  const QuadLane q = quad_lane();
  const u32 lane = threadIdx.x & 63;
  const u32 groups = uniform(pe.a[5]);
  // groups indexed -> 256-row spans whose groups are both indexed (the last span of a stream with an
  // odd number of groups has one)
  auto spans = [&](u32 g) { return g >= groups ? (groups + 1) >> 1 : g >> 1; };
  ring_pass<true, kStreamWaves, 6>(
      c, pe, be, jump_off, [](const ChainState& st) { return st.g; },
      [&](u32 g_lo, u32 g_hi, u32 uw) {
        for (u32 P = spans(g_lo) + uw; P < spans(g_hi); P += kUnpackWaves) {
          const u32 G = 2 * P + q.grp;
          const bool have = G < groups;
          const u64 ee = lds64(be.index_off + 8 * (have ? G : 2 * P));
          const u32 first = (u32)ee, hdr = (u32)(ee >> 32);
          const u32 b = __builtin_amdgcn_ubfe(hdr, q.sh, 8);
          // a span whose miniblocks are all narrower than the exception value holds no patch row
          if (__builtin_amdgcn_ballot_w64(have && b >= xc.need_bits) == 0) continue;
          const u32 w = __builtin_amdgcn_sad_u8(hdr & q.pm, 0u, first);
          u32 f = 0;
          if (__builtin_amdgcn_ballot_w64(!widths_le8(hdr)) == 0) {
            const u32 bit = __umul24(q.k4, b);
            const u32x2 pair = ring_pair(((w + (bit >> 5)) << 2) & (kStreamRingBytes - 1));
            const u32 x = __builtin_amdgcn_alignbit(pair.y, pair.x, bit);
#pragma unroll
            for (u32 k = 0; k < 4; k++) f |= (__builtin_amdgcn_ubfe(x, k * b, b) == xc.exception ? 1u : 0u) << k;
          } else {  // streams no compressor writes (exponents are below 32): one value at a time
#pragma unroll
            for (u32 k = 0; k < 4; k++) {
              const u32 bit = __umul24(q.k4 + k, b);
              const u32x2 pair = ring_pair(((w + (bit >> 5)) << 2) & (kStreamRingBytes - 1));
              f |= (take_bits(pair.x, pair.y, bit, b) == xc.exception ? 1u : 0u) << k;
            }
          }
          const u32 row = 256 * P + 4 * lane;  // a multiple of four: the flags stay inside one 32-bit word
          if (have && f && (row >> 6) < words)
            atomicOr(reinterpret_cast<u32*>(lds_raw + rk.bits_off + 4 * (row >> 5)), f << (row & 31));
        }
      });
  // rows of the VariableByte tail (decoded by the walker at the end of the pass)
  if (threadIdx.x < 32) {
    const u32 i = be.rounded + threadIdx.x;
    if (i < n && lds32(be.tail_off + 4 * threadIdx.x) == xc.exception) bitmap_or(rk.bits_off, i);
  }
  __syncthreads();
}

__device__ __forceinline__ void stream_pseudo(const Ctx& c, const Pipe& root, const Pipe& pd, const Pipe& pe,
                                              const RankView& rk, u64* __restrict__ dst, u32 frac_off, u32 jump_off,
                                              const ExpCode& xc) {
  const u32 lane = threadIdx.x & 63;
  const BpView bd = resolve_bp(c, pd), be = resolve_bp(c, pe);
  const Tab patches = resolve_tab(c, uniform(root.a[2]));
  const u32 n = uniform(root.count);
  const u32 words = n / 64 + 1;
  const __amdgpu_buffer_rsrc_t rsrc_d = make_rsrc(bd.base, bd.bytes), rsrc_e = make_rsrc(be.base, be.bytes);
  const __amdgpu_buffer_rsrc_t dst_rsrc = make_rsrc(dst, n * 8);
  u32 exbad = 0;
  u32 row_end = 0;  // rows the pass could produce (walker wave only; shared through LDS below)

  u32 idx_max = 0;
  const u32 groups_d = uniform(pd.a[5]);
  // patch candidates are loaded through a descriptor: an index past the last patch reads 0
  const __amdgpu_buffer_rsrc_t rsrc_p = make_rsrc(patches.g, patches.count * 8);
  // Digit number d out of the ring: its group's index entry gives the first payload word and the
  // four widths.
  auto ring_digit = [&](u32 d) -> u32 {
    const u64 ed = lds64(bd.index_off + 8 * (d >> 7));
    const u32 m = (d >> 5) & 3, hdr = (u32)(ed >> 32);
    const u32 b = __builtin_amdgcn_ubfe(hdr, 24 - 8 * m, 8);
    const u32 w = __builtin_amdgcn_sad_u8(hdr & ~(u32)(0xFFFFFFFFull >> (8 * m)), 0u, (u32)ed);
    const u32 bit = __umul24(d & 31, b);
    const u32x2 pair = ring_pair(((w + (bit >> 5)) << 2) & (kStreamRingBytes - 1));
    return take_bits(pair.x, pair.y, bit, b);
  };
  // Lane l produces two pairs of consecutive rows of a 256-row span P: rows 256P + 2l, + 1 (pair A,
  // in group 2P of the exponent stream) and rows 256P + 128 + 2l, + 1 (pair B, in group 2P + 1). A
  // pair is 16 bytes of output, so each of the span's two store instructions writes 1 KiB of
  // consecutive bytes (whole cache lines). Both pairs sit in miniblock l / 16 of their groups, at
  // positions 2 * (l % 16).
  const u32 pm_sh = 24 - 8 * (lane >> 4);
  const u32 pm_mask = (lane >> 4) ? ~(0xFFFFFFFFu >> (8 * (lane >> 4))) : 0u;
  const u32 k2 = 2 * (lane & 15);
  // bits of a 64-bit bitmap word below the pair's first row; the pair's two flags sit at bit 2l % 64
  const u32 below_lo = (2 * lane & 63) < 32 ? (1u << (2 * lane & 31)) - 1 : ~0u;
  const u32 below_hi = (2 * lane & 63) < 32 ? 0u : (1u << (2 * lane & 31)) - 1;
  // Exponent words of a wave's next span, in flight while the current span is produced: they come
  // from global memory, and their latency would otherwise be paid once per span. With them: the
  // exceptions in front of each pair and the pairs' exception flags.
  // The exponent words themselves are fetched once per wave, not once per lane: the span's two groups
  // are at most 42 consecutive words when every width is at most 5, lane j loads word j of them (one
  // coalesced load) and the lanes pick their two words with lane permutes. Sixty-four lanes that each
  // load their own, mostly identical, 8 bytes cost the vector cache a pass per lane: 8 % of the
  // launch for plain exponents and 14 % for dictionary codes, whose lanes all hit the same word.
  struct PairFetch {
    u32 eb, bit, word, xb;
  };
  PairFetch pf[2];
  u32 pf_span = NONE, pf_f = 0, pf_words = 0, pf_first = 0;
  bool pf_narrow = false;
#pragma unroll
  for (u32 h = 0; h < 2; h++) pf[h].eb = pf[h].bit = pf[h].word = pf[h].xb = 0;
  const u32 spans_e = be.rounded >> 8;
  auto exp_fetch = [&](u32 P) {
    pf_span = P;
    const u32 Ps = P < spans_e ? P : 0;
    const u32 w64 = 4 * Ps + (lane >> 5);
    u32 f = 0;
    bool narrow = true;
#pragma unroll
    for (u32 h = 0; h < 2; h++) {
      const u64 word = lds64(rk.bits_off + 8 * (w64 + 2 * h));
      const u32 wl = (u32)word, wh = (u32)(word >> 32);
      pf[h].xb = lds32(rk.rank_off + 4 * (w64 + 2 * h)) + (u32)__builtin_popcount(wl & below_lo) +
                 (u32)__builtin_popcount(wh & below_hi);
      f |= __builtin_amdgcn_ubfe((lane & 16) ? wh : wl, 2 * lane & 31, 2) << (2 * h);
      const u64 ee = lds64(be.index_off + 8 * (2 * Ps + h));  // the same entry for every lane
      const u32 ef = uniform((u32)ee), eh = uniform((u32)(ee >> 32));
      narrow = narrow && widths_le5(eh);
      pf[h].eb = __builtin_amdgcn_ubfe(eh, pm_sh, 8);
      pf[h].word = __builtin_amdgcn_sad_u8(eh & pm_mask, 0u, ef);
      pf[h].bit = __umul24(k2, pf[h].eb);
      if (h == 0) {
        pf_first = ef - 1;  // the header of the span's first group
        pf_words = __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, (pf_first + lane) * 4, 0, 0);  // past the stream: 0
      }
    }
    pf_f = f;
    pf_narrow = narrow;
  };
  // rows lag up to one 256-row span behind the digits: their words may sit in the previous batch
  u32 iteration = 0;  // batches this unpack wave has seen
  ring_pass<true, kStreamWaves, 9, true, true>(
      c, pd, bd, jump_off,
      [&](const ChainState& st) {
        // digits indexed so far -> rows that can be produced, in whole 256-row spans
        const u32 digits_ready = st.g * btrfmt::FBP_GROUP < bd.rounded ? st.g * btrfmt::FBP_GROUP : bd.rounded;
        u32 r = select_clear(rk, words, n, digits_ready);
        r = r < be.rounded ? r : be.rounded;
        r &= ~255u;
        row_end = r;
        return r;
      },
      [&](u32 r_lo, u32 r_hi, u32 uw) {
        // A wave takes a 256-row span (two groups of the exponent stream) per iteration. Per pair the
        // two exponents come out of one 8-byte load and one funnel shift (they are at most 5 bits
        // wide), one bitmap word and one rank entry give the exceptions in front of it, one 16-byte
        // load its (at most two) patches, and the pair leaves as one 16-byte store.
        // Span P belongs to unpack wave P % kUnpackWaves in every batch, so a wave knows its next span
        // and loads that span's exponent words while it works on the current one: the global load
        // latency would otherwise be paid once per batch, between two barriers.
        iteration++;
#ifdef BTR_DEBUG_STEPS
        if (c.debug & 16) {  // diagnostic: loader, walker and barriers only
          if (uw == 0 && lane == 0) lds_st32(ring_release_word(c), iteration);
          return;
        }
#endif
        u32 P = r_lo >> 8;
        P += (uw + kUnpackWaves - P % kUnpackWaves) % kUnpackWaves;
        for (; P < (r_hi >> 8); P += kUnpackWaves) {
          // the span's first digit group (from the exceptions in front of the span) and the index entries
          // of the three groups it can touch: two dependent LDS reads, started before anything else
          const u32 xb0 = uniform(lds32(rk.rank_off + 16 * P));
          u32 g0 = (256 * P - xb0) >> 7;
          g0 = g0 < groups_d ? g0 : groups_d - 1;
          const u64 en = lds64(bd.index_off + 8 * (g0 + (lane < 3 ? lane : 0)));  // lanes 0..2: the three entries
          if (pf_span != P) exp_fetch(P);  // first span of this wave
          const PairFetch cur[2] = {pf[0], pf[1]};
          const u32 f = pf_f, span_words = pf_words, span_first = pf_first;
          const bool narrow = pf_narrow;
          // patch candidates: loaded by the (few) lanes whose pair holds an exception row, through a
          // descriptor that returns 0 past the last patch, and consumed right before the stores. A load
          // by all 64 lanes of what is almost always the same 16 bytes costs the vector cache a pass per lane.
          u32x4 pt[2];
#pragma unroll
          for (u32 h = 0; h < 2; h++) {
            pt[h].x = pt[h].y = pt[h].z = pt[h].w = 0;
            if ((f >> (2 * h)) & 3) pt[h] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_p, cur[h].xb * 8, 0, 0);
          }
          exp_fetch(P + kUnpackWaves);
          u32 ex[4];  // rows A0, A1, B0, B1
          if (narrow) {
#pragma unroll
            for (u32 h = 0; h < 2; h++) {
              const u32 rel = cur[h].word + (cur[h].bit >> 5) - span_first;  // below 43
              const u32 lo = __builtin_amdgcn_ds_bpermute(rel << 2, span_words);
              const u32 hi = __builtin_amdgcn_ds_bpermute((rel + 1) << 2, span_words);
              const u32 x = __builtin_amdgcn_alignbit(hi, lo, cur[h].bit);
              ex[2 * h] = __builtin_amdgcn_ubfe(x, 0u, cur[h].eb);
              ex[2 * h + 1] = __builtin_amdgcn_ubfe(x, cur[h].eb, cur[h].eb);
            }
          } else {  // streams no compressor writes: one exponent at a time
#pragma unroll
            for (u32 k = 0; k < 4; k++) {
              const PairFetch& pr = cur[k >> 1];
              const u32 ebit = __umul24(k2 + (k & 1), pr.eb);
              const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrc_e, (pr.word + (ebit >> 5)) * 4, 0, 0);
              ex[k] = take_bits(t.x, t.y, ebit, pr.eb);
            }
          }
          // digits: a row takes digit number (row - exceptions before it). The span's digits lie in at
          // most three groups of the digit stream, the first one known from the exceptions in front of
          // the span. Straight path: those groups are packed with one width below 32, so digit d sits at
          // bit (d - first digit of the first group) * width, plus one header word per group boundary
          // crossed.
          u32 dv[4];
          {
            const u32 hh = (u32)(en >> 32);
            const u32 h0 = __builtin_amdgcn_readlane(hh, 0), h1 = __builtin_amdgcn_readlane(hh, 1),
                      h2 = __builtin_amdgcn_readlane(hh, 2), first0 = __builtin_amdgcn_readlane((u32)en, 0);
            const u32 bw = h0 & 0xFF;
            const bool straight = h0 == bw * 0x01010101u && bw < 32 && (g0 + 1 >= groups_d || h1 == h0) &&
                                  (g0 + 2 >= groups_d || h2 == h0);
            u32 dg[4];  // digit numbers of rows A0, A1, B0, B1
#pragma unroll
            for (u32 h = 0; h < 2; h++) {
              dg[2 * h] = 256 * P + 128 * h + 2 * lane - cur[h].xb;  // digit of the pair's first row
              dg[2 * h + 1] = dg[2 * h] + 1 - ((f >> (2 * h)) & 1);
            }
            if (straight) {
              // all four ring reads are issued before the first one is waited for
              u32 pos[4];
              u32x2 pair[4];
#pragma unroll
              for (u32 k = 0; k < 4; k++) {
                const u32 rel = dg[k] - 128 * g0;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(pos[k]) : "v"(rel), "s"(bw), "v"((rel >> 7) << 5));
              }
#pragma unroll
              for (u32 k = 0; k < 4; k++) pair[k] = ring_pair(((first0 + (pos[k] >> 5)) << 2) & (kStreamRingBytes - 1));
#pragma unroll
              for (u32 k = 0; k < 4; k++)
                dv[k] = __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(pair[k].y, pair[k].x, pos[k]), 0u, bw);
            } else {  // beyond the last digit (patch rows only): any LDS word will do
#pragma unroll
              for (u32 k = 0; k < 4; k++) dv[k] = ring_digit(dg[k]);
            }
          }
          if (P == (r_lo >> 8)) {
            // the first span of the iteration is the only one whose digits may sit in the previous
            // batch: its words are in registers now, the loader may refill that slot
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_st32(ring_release_word(c), iteration);
          }
          u64 out[4];
#pragma unroll
          for (u32 k = 0; k < 4; k++) {
            // narrow exponent groups hold values below 32 only: table index as it is
            const u32 idx = narrow ? ex[k] : ex[k] & btrfmt::DEC_INDEX_MASK;
            idx_max = idx_max > ex[k] ? idx_max : ex[k];
            const u64 fr = lds64(frac_off + 8 * idx);
            out[k] = (u64)__double_as_longlong((double)(int)dv[k] * __longlong_as_double((long long)fr));
          }
#pragma unroll
          for (u32 h = 0; h < 2; h++) {
            // the pair's first exception takes the first candidate, a second one the second
            const u64 p0 = ((u64)pt[h].y << 32) | pt[h].x, p1 = ((u64)pt[h].w << 32) | pt[h].z;
            const bool first_is_patch = ((f >> (2 * h)) & 1) != 0;
            if (ex[2 * h] == xc.exception) out[2 * h] = p0;
            if (ex[2 * h + 1] == xc.exception) out[2 * h + 1] = first_is_patch ? p1 : p0;
          }
          u32x4 s0, s1;
          s0.x = (u32)out[0], s0.y = (u32)(out[0] >> 32), s0.z = (u32)out[1], s0.w = (u32)(out[1] >> 32);
          s1.x = (u32)out[2], s1.y = (u32)(out[2] >> 32), s1.z = (u32)out[3], s1.w = (u32)(out[3] >> 32);
#ifdef BTR_DEBUG_STEPS
          if ((c.debug & 32) && s0.x != 0x12345678u) continue;  // diagnostic: no stores
#endif
          const u32 o = 2048 * P + 16 * lane;
          __builtin_amdgcn_raw_buffer_store_b128(s0, dst_rsrc, o, 0, kStoreNt);
          __builtin_amdgcn_raw_buffer_store_b128(s1, dst_rsrc, o + 1024, 0, kStoreNt);
        }
      });
  // plain exponents: values above 23 are no table index; codes: values outside the dictionary
  u32 codebad = 0;
  if (xc.codes ? idx_max >= xc.codes : idx_max > btrfmt::DEC_EXCEPTION_CODE) (xc.codes ? codebad : exbad) |= 2;
  // rows behind the last one the pass could produce: digits or exponents from the tails
  const u32 end_off = c.red_off + 32;
  if (threadIdx.x == 0) lds_st32(end_off, row_end);
  __syncthreads();
  row_end = uniform(lds32(end_off));
  for (u32 i = row_end + threadIdx.x; i < n; i += kStreamThreads) {
    u32 x;
    if (i >= be.rounded) {
      x = lds32(be.tail_off + 4 * ((i - be.rounded) & 31));
    } else {
      const u64 ee = lds64(be.index_off + 8 * (i >> 7));
      const u32 m = (i >> 5) & 3;
      x = global_extract(rsrc_e, (u32)ee, (u32)(ee >> 32), 24 - 8 * m, ~(0xFFFFFFFFu >> (8 * m)), i & 31);
    }
    bool member;
    const u32 xr = rank_lt(rk, i, &member);
    const u32 di = i - xr;
    u32 dv = 0;
    if (x != xc.exception) {
      if (di >= bd.rounded) {
        dv = lds32(bd.tail_off + 4 * ((di - bd.rounded) & 31));
      } else {
        const u64 ed = lds64(bd.index_off + 8 * (di >> 7));
        const u32 m = (di >> 5) & 3;
        dv = global_extract(rsrc_d, (u32)ed, (u32)(ed >> 32), 24 - 8 * m, ~(0xFFFFFFFFu >> (8 * m)), di & 31);
      }
    }
    u32 idx = x & btrfmt::DEC_INDEX_MASK;
    if (xc.codes) {
      codebad |= x >= xc.codes ? 2u : 0u;
    } else {
      exbad |= (x != xc.exception && idx > btrfmt::DEC_MAX_EXPONENT) ? 2u : 0u;
      idx = idx > btrfmt::DEC_MAX_EXPONENT ? btrfmt::DEC_MAX_EXPONENT : idx;
    }
    u64 r = (u64)__double_as_longlong((double)(int)dv * __longlong_as_double((long long)lds64(frac_off + 8 * idx)));
    if (x == xc.exception) r = xr < patches.count ? ldg64(patches.g + 8ull * xr) : 0;
    dst[i] = r;
  }
  if (codebad) raise(c, ST_DICT_CODE);
  if (exbad & 2) raise(c, ST_EXPONENT);
}

template <u32 BASE, u32 IA, u32 IB, u32 IPS>
__device__ __forceinline__ void mat_dbl(const Ctx& c, const DblEval& e, const Tab& out, u32 count) {
  const u32 tid = threadIdx.x;
  u32 bad = 0;
  for (u32 base = 0; base < count; base += kBlockThreads * kUnroll) {
    u64 v[kUnroll];
#pragma unroll
    for (u32 u = 0; u < kUnroll; u++) {
      const u32 i = base + u * kBlockThreads + tid;
      v[u] = eval_dbl<BASE, IA, IB, IPS>(e, i < count ? i : count - 1, bad);
    }
#pragma unroll
    for (u32 u = 0; u < kUnroll; u++) {
      const u32 i = base + u * kBlockThreads + tid;
      if (i < count) tab_st64(out, i, v[u]);
    }
  }
  report(c, bad);
}

// Two consecutive rows per lane (16-byte stores) for table-driven double roots, the counterpart
// of mat_int4.
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

template <u32 BASE, bool TAB_LDS = false>
__device__ __forceinline__ u64x2 eval_dbl2_base(const DblEval& e, u32 i, __amdgpu_buffer_rsrc_t seq_rsrc) {
  u64x2 v;
  if (BASE == B_ONE) {
    v.x = v.y = ((u64)e.k1 << 32) | e.k0;
  } else if (BASE == B_SEQ) {
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(seq_rsrc, i * 8, 0, 0);
    v.x = ((u64)w.y << 32) | w.x;
    v.y = ((u64)w.w << 32) | w.z;
  } else if (BASE == B_RLE) {
    const u32 w = i >> 6, p = i & 63;
    const u64 word = lds64(e.rk.bits_off + 8 * w);
    const u32 r = lds32(e.rk.rank_off + 4 * w) + __popcll(word & ((2ull << p) - 1));
    const u32 last = e.tab.count - 1;
    u32 r0 = r ? r - 1 : 0;
    u32 r1 = r0 + ((u32)(word >> (p + 1)) & 1);
    r0 = r0 < last ? r0 : last;
    r1 = r1 < last ? r1 : last;
    v.x = TAB_LDS ? lds64(e.tab.lds + 8 * r0) : tab_ld64(e.tab, r0);
    v.y = TAB_LDS ? lds64(e.tab.lds + 8 * r1) : tab_ld64(e.tab, r1);
  } else {  // B_FREQ
    const u32 w = i >> 6, p = i & 63;
    const u64 word = lds64(e.rk.bits_off + 8 * w);
    const u32 m = (u32)(word >> p) & 3;
    v.x = v.y = ((u64)e.k1 << 32) | e.k0;
    if (m) {
      const u32 last = e.tab.count - 1;
      u32 r = lds32(e.rk.rank_off + 4 * w) + __popcll(word & ((1ull << p) - 1));
      if (m & 1) v.x = tab_ld64(e.tab, r < last ? r : last);
      r += m & 1;
      if (m & 2) v.y = tab_ld64(e.tab, r < last ? r : last);
    }
  }
  return v;
}

template <u32 BASE, bool TAB_LDS = false>
__device__ __forceinline__ void mat_dbl2(const Ctx& c, const DblEval& e, u64* __restrict__ dst, u32 count) {
  const u32 tid = threadIdx.x;
  const u32 pairs = count & ~1u;
  const __amdgpu_buffer_rsrc_t seq_rsrc =
      BASE == B_SEQ && e.tab.loc != LOC_LDS ? make_rsrc(e.tab.g, e.tab.count * 8) : make_rsrc(nullptr, 0);
  constexpr u32 kUnrollHere = BASE == B_SEQ ? 2 * kUnroll4 : kUnroll4;  // copies: more loads in flight
  constexpr u32 kRowsPerIter = kBlockThreads * 2 * kUnrollHere;
  const u32 whole = pairs / kRowsPerIter * kRowsPerIter;  // whole iterations: no clamps, no store predicates
  auto iteration = [&](u32 base, auto tag) {
    constexpr bool PARTIAL = decltype(tag)::value;
    u64x2 v[kUnrollHere];
#pragma unroll
    for (u32 u = 0; u < kUnrollHere; u++) {
      const u32 i = base + (u * kBlockThreads + tid) * 2;
      v[u] = eval_dbl2_base<BASE, TAB_LDS>(e, PARTIAL ? (i < pairs ? i : pairs - 2) : i, seq_rsrc);
    }
#pragma unroll
    for (u32 u = 0; u < kUnrollHere; u++) {
      const u32 i = base + (u * kBlockThreads + tid) * 2;
      if (!PARTIAL || i < pairs) BTR_NT_STORE(v[u], reinterpret_cast<u64x2*>(dst + i));
    }
  };
  for (u32 base = 0; base < whole; base += kRowsPerIter) iteration(base, std::false_type{});
  if (whole < pairs) iteration(whole, std::true_type{});
  u32 bad = 0;
  if (tid < count - pairs) dst[pairs + tid] = eval_dbl<BASE, GEN, GEN, PS_GEN>(e, pairs + tid, bad);
}

// PSEUDODECIMAL whose digits and exponents are both run-length encoded (bare RLE pipes), four
// consecutive rows per lane: the rows share one word of each bitmap (run starts of the exponents,
// exception flags, and - up to a word boundary - run starts of the digits), so each of the three
// rank lookups is made once and advanced by flag bits for the other rows. Results are those of
// eval_dbl<B_PSEUDO> row by row (double/Pseudodecimal.cpp:424-436).
// EXP_LDS / DIG_LDS: where the run-value tables sit, fixed at compile time so that a lane's loads
// are not separated by location tests.
template <bool EXP_LDS, bool DIG_LDS>
__device__ __forceinline__ void mat_pseudo_rle4(const Ctx& c, const DblEval& e, u64* __restrict__ dst, u32 count) {
  const u32 tid = threadIdx.x;
  const u32 quads = count & ~3u;
  u32 bad = 0;
  const u32 e_last = e.ib.tab.count - 1, d_last = e.k0 ? e.ia.tab.count - 1 : 0;
  const u32 di_last = e.k0 ? e.k0 - 1 : 0;
  const bool patches_lds = e.tab.loc == LOC_LDS;
  const __amdgpu_buffer_rsrc_t rsrc_p = make_rsrc(patches_lds ? nullptr : e.tab.g, patches_lds ? 0 : e.tab.count * 8);
  // two quads per lane and iteration: the table loads of both (the digit values may sit in the
  // chunk's scratch slice in global memory) are in flight together
  auto quad = [&](u32 first, u64 (&out)[4]) -> u32 {
    const u32 i = first < quads ? first : quads - 4;  // lanes past the end repeat the last quad
    const u32 w = i >> 6, p = i & 63;
    // exponents: run index of row i + k = starts at or before it, minus one
    const u64 se = lds64(e.ib.rk.bits_off + 8 * w);
    const u32 ne = lds32(e.ib.rk.rank_off + 4 * w) + (u32)__popcll(se & ((2ull << p) - 1));
    const u32 fe = (u32)((se >> p) >> 1);  // start flags of rows i+1..
    // exceptions strictly before row i + k
    const u64 sx = lds64(e.rk.bits_off + 8 * w);
    const u32 nx = lds32(e.rk.rank_off + 4 * w) + (u32)__popcll(sx & ((1ull << p) - 1));
    const u32 fx = (u32)(sx >> p) & 15;
    u32 ex[4], xr[4], di[4];
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
      const u32 n = ne + (u32)__builtin_popcount(fe & ((1u << k) - 1));
      u32 r = n ? n - 1 : 0;
      r = r < e_last ? r : e_last;
      ex[k] = EXP_LDS ? lds32(e.ib.tab.lds + 4 * r) : ldg32(e.ib.tab.g + 4ull * r);
      xr[k] = nx + (u32)__builtin_popcount(fx & ((1u << k) - 1));
      const u32 d = i + k - xr[k];
      di[k] = d < di_last ? d : di_last;
    }
    // patch candidates for all four rows, loaded before anything waits for them (an index past the
    // last patch reads 0 through the descriptor); a row takes its candidate if its exponent says so
    u64 pw[4];
    if (patches_lds) {
#pragma unroll
      for (u32 k = 0; k < 4; k++) pw[k] = xr[k] < e.tab.count ? lds64(e.tab.lds + 8 * xr[k]) : 0;
    } else {
#pragma unroll
      for (u32 k = 0; k < 4; k++) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrc_p, xr[k] * 8, 0, 0);
        pw[k] = ((u64)t.y << 32) | t.x;
      }
    }
    // digits: rank lookup for the first digit index, flag bits for the (at most three) following ones
    u32 dg[4] = {0, 0, 0, 0};
    if (e.k0) {
      const u32 wd = di[0] >> 6, pd = di[0] & 63;
      const u64 s0 = lds64(e.ia.rk.bits_off + 8 * wd), s1 = lds64(e.ia.rk.bits_off + 8 * wd + 8);
      const u32 nd = lds32(e.ia.rk.rank_off + 4 * wd) + (u32)__popcll(s0 & ((2ull << pd) - 1));
      const u32 fd = (u32)(((s0 >> pd) >> 1) | ((s1 << (63 - pd))));  // start flags of digits di[0]+1..
#pragma unroll
      for (u32 k = 0; k < 4; k++) {
        const u32 ahead = di[k] - di[0];  // 0..3
        const u32 n = nd + (u32)__builtin_popcount(fd & ((1u << ahead) - 1));
        u32 r = n ? n - 1 : 0;
        r = r < d_last ? r : d_last;
        dg[k] = DIG_LDS ? lds32(e.ia.tab.lds + 4 * r) : ldg32(e.ia.tab.g + 4ull * r);
      }
    }
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
      u32 idx = ex[k] & btrfmt::DEC_INDEX_MASK;
      bad |= (ex[k] != btrfmt::DEC_EXCEPTION_CODE && idx > btrfmt::DEC_MAX_EXPONENT) ? 2u : 0u;
      idx = idx > btrfmt::DEC_MAX_EXPONENT ? btrfmt::DEC_MAX_EXPONENT : idx;
      const u64 fbits = ((u64)(u32)__builtin_amdgcn_ds_bpermute(idx << 2, e.frac_hi) << 32) |
                        (u32)__builtin_amdgcn_ds_bpermute(idx << 2, e.frac_lo);
      out[k] = (u64)__double_as_longlong((double)(int)dg[k] * __longlong_as_double((long long)fbits));
      if (ex[k] == btrfmt::DEC_EXCEPTION_CODE) out[k] = pw[k];
    }
    return i;
  };
  auto put = [&](u32 first, u32 i, const u64 (&out)[4]) {
    if (first < quads) {
      u64x2 lo, hi;
      lo.x = out[0], lo.y = out[1], hi.x = out[2], hi.y = out[3];
      BTR_NT_STORE(lo, reinterpret_cast<u64x2*>(dst + i));
      BTR_NT_STORE(hi, reinterpret_cast<u64x2*>(dst + i + 2));
    }
  };
  for (u32 base = 0; base < quads; base += kBlockThreads * 8) {
    u64 oa[4], ob[4];
    const u32 fa = base + tid * 4, fb = fa + kBlockThreads * 4;
    const u32 ia = quad(fa, oa), ib = quad(fb, ob);
    put(fa, ia, oa);
    put(fb, ib, ob);
  }
  {
    // the last count % 4 rows; every lane evaluates (the powers of ten are fetched by lane permute,
    // which reads from active lanes only), the first few store
    const u64 v = eval_dbl<B_PSEUDO, B_RLE, B_RLE, PS_NONE>(e, quads + tid < count ? quads + tid : count - 1, bad);
    if (tid < count - quads) dst[quads + tid] = v;
  }
  report(c, bad);
}

// Whether a double root can take the two-rows-per-lane path into a 16-byte aligned output.
__device__ __forceinline__ bool wide_rows_dbl(const DblEval& e, const Tab& out, u32 count) {
  if (out.loc == LOC_LDS || (reinterpret_cast<u64>(out.g) & 15) != 0 || count < 2) return false;
  if (e.base == B_SEQ) return e.tab.loc != LOC_LDS && (reinterpret_cast<u64>(e.tab.g) & 3) == 0;
  if (e.base == B_RLE) return e.tab.count > 0;
  return e.base == B_ONE || e.base == B_FREQ;
}

template <u32 BASE>
__device__ __forceinline__ void mat_int_ps(const Ctx& c, const IntEval& e, const Tab& out, u32 count) {
  switch (post_signature(e)) {
    case PS_NONE: mat_int<BASE, PS_NONE>(c, e, out, count); break;
    case PS_ADD: mat_int<BASE, PS_ADD>(c, e, out, count); break;
    case PS_DICT_LDS: mat_int<BASE, PS_DICT_LDS>(c, e, out, count); break;
    default: mat_int<BASE, PS_GEN>(c, e, out, count); break;
  }
}

template <bool DOUBLE_KERNEL>
__device__ __forceinline__ void materialize(const Ctx& c, u32 pipe_id, u32 seq_id) {
  const u32 count = uniform(pipe_at(c, pipe_id).count);
  const bool is_double = uniform(pipe_at(c, pipe_id).is_double) != 0;
  const Tab out = resolve_tab(c, seq_id);
  if (count == 0) return;
  if (!is_double) {
    const IntEval e = resolve_int(c, pipe_id);
    if (wide_rows(e, out, count)) {
      switch (e.base) {
        case B_SEQ: mat_int4_ps<B_SEQ>(c, e, out, count); break;
        case B_ONE: mat_int4_ps<B_ONE>(c, e, out, count); break;
        case B_RLE: mat_int4_ps<B_RLE>(c, e, out, count); break;
        default: mat_int4_ps<B_FREQ>(c, e, out, count); break;
      }
      return;
    }
    switch (e.base) {
      case B_SEQ: mat_int_ps<B_SEQ>(c, e, out, count); break;
      case B_ONE: mat_int<B_ONE, PS_GEN>(c, e, out, count); break;
      case B_BP:
        switch (post_signature(e)) {
          case PS_NONE: mat_int_bp<PS_NONE>(c, e, out, count); break;
          case PS_ADD: mat_int_bp<PS_ADD>(c, e, out, count); break;
          case PS_DICT_LDS: mat_int_bp<PS_DICT_LDS>(c, e, out, count); break;
          default: mat_int_bp<PS_GEN>(c, e, out, count); break;
        }
        break;
      case B_RLE: mat_int_ps<B_RLE>(c, e, out, count); break;
      case B_FREQ: mat_int<B_FREQ, PS_GEN>(c, e, out, count); break;
      default: break;
    }
  } else if (DOUBLE_KERNEL) {
    const DblEval e = resolve_dbl(c, pipe_id);
    if (wide_rows_dbl(e, out, count)) {
      u64* dst = reinterpret_cast<u64*>(const_cast<u8*>(out.g));
      switch (e.base) {
        case B_SEQ: mat_dbl2<B_SEQ>(c, e, dst, count); break;
        case B_ONE: mat_dbl2<B_ONE>(c, e, dst, count); break;
        case B_RLE:
          if (e.tab.loc == LOC_LDS)
            mat_dbl2<B_RLE, true>(c, e, dst, count);
          else
            mat_dbl2<B_RLE>(c, e, dst, count);
          break;
        default: mat_dbl2<B_FREQ>(c, e, dst, count); break;
      }
      return;
    }
    const bool bare = e.ia.n_post == 0 && e.ib.n_post == 0;
    switch (e.base) {
      case B_SEQ: mat_dbl<B_SEQ, GEN, GEN, PS_GEN>(c, e, out, count); break;
      case B_ONE: mat_dbl<B_ONE, GEN, GEN, PS_GEN>(c, e, out, count); break;
      case B_RLE: mat_dbl<B_RLE, GEN, GEN, PS_GEN>(c, e, out, count); break;
      case B_FREQ: mat_dbl<B_FREQ, GEN, GEN, PS_GEN>(c, e, out, count); break;
      case B_DICTD:
        if (e.ia.base == B_BP && bare)
          mat_dbl<B_DICTD, B_BP, GEN, PS_NONE>(c, e, out, count);
        else
          mat_dbl<B_DICTD, GEN, GEN, PS_GEN>(c, e, out, count);
        break;
      case B_PSEUDO:
        if (e.ib.base == B_BP && (e.ia.base == B_BP || e.k0 == 0) && bare)
          mat_dbl<B_PSEUDO, B_BP, B_BP, PS_NONE>(c, e, out, count);
        else if (e.ib.base == B_RLE && (e.ia.base == B_RLE || e.k0 == 0) && bare) {
          if (out.loc != LOC_LDS && (reinterpret_cast<u64>(out.g) & 15) == 0 && count >= 4 && e.ib.tab.count > 0 &&
              (e.k0 == 0 || e.ia.tab.count > 0))
          {
            u64* d = reinterpret_cast<u64*>(const_cast<u8*>(out.g));
            const bool el = e.ib.tab.loc == LOC_LDS, dl = e.k0 == 0 || e.ia.tab.loc == LOC_LDS;
            if (el && dl)
              mat_pseudo_rle4<true, true>(c, e, d, count);
            else if (el)
              mat_pseudo_rle4<true, false>(c, e, d, count);
            else if (dl)
              mat_pseudo_rle4<false, true>(c, e, d, count);
            else
              mat_pseudo_rle4<false, false>(c, e, d, count);
          }
          else
            mat_dbl<B_PSEUDO, B_RLE, B_RLE, PS_NONE>(c, e, out, count);
        }
        else if (e.ib.base == B_RLE && e.ia.base == B_BP && bare)
          mat_dbl<B_PSEUDO, B_BP, B_RLE, PS_NONE>(c, e, out, count);
        else
          mat_dbl<B_PSEUDO, GEN, GEN, PS_GEN>(c, e, out, count);
        break;
      case B_DBP: mat_dbl<B_DBP, B_BP, GEN, PS_NONE>(c, e, out, count); break;
      default: break;
    }
  }
}

// ---- bitmap steps ----------------------------------------------------------------------------------

template <u32 THREADS = kBlockThreads>
__device__ __forceinline__ void bitmap_clear(const Ctx& c, const Step& s) {
  const Seq& bm = seq_at(c, uniform(s.a[1]));
  const u32 off = c.table_base + uniform(bm.off), words = uniform(bm.count) * 2;
  for (u32 w = threadIdx.x; w < words; w += THREADS) lds_st32(off + 4 * w, 0);
}

// Members of one CRoaring container (or of the whole bitmap for the u32-array form) -> bits.
template <u32 THREADS = kBlockThreads>
__device__ __forceinline__ void bitmap_set(const Ctx& c, const Step& s) {
  const u8* src = c.src + uniform(s.a[0]);
  const u32 bits_off = seq_lds(c, uniform(s.a[1]));
  const u32 count = uniform(s.a[2]), first = uniform(s.a[3]), nbits = uniform(s.a[4]), kind = uniform(s.a[5]);
  u32 bad = 0;
  for (u32 k = threadIdx.x; k < count; k += THREADS) {
    switch (kind) {
      case SET_ARRAY32: {
        const u32 r = ldg32(src + 4ull * k);
        if (r < nbits)
          bitmap_or(bits_off, r);
        else
          bad = 1;
        break;
      }
      case SET_ARRAY16: {
        const u32 r = first + ldg16(src + 2ull * k);
        if (r < nbits)
          bitmap_or(bits_off, r);
        else
          bad = 1;
        break;
      }
      case SET_BITSET: {  // k indexes the container's 1024 u64 words
        const u64 word = ldg64(src + 8ull * k);
        const u32 row = first + 64 * k;
        if (word) {
          if (row + 63 < nbits || (row < nbits && (word >> (nbits - row)) == 0)) {
            atomicOr(reinterpret_cast<u32*>(lds_raw + bits_off + 8 * (row >> 6)), (u32)word);
            atomicOr(reinterpret_cast<u32*>(lds_raw + bits_off + 8 * (row >> 6) + 4), (u32)(word >> 32));
          } else {
            bad = 1;
          }
        }
        break;
      }
      default: {  // SET_RUNS: {u16 start, u16 length-1}
        const u32 start = first + ldg16(src + 4ull * k);
        const u32 last = start + ldg16(src + 4ull * k + 2);
        if (last >= nbits) {
          bad = 1;
          break;
        }
        for (u32 w = start >> 5; w <= (last >> 5); w++) {
          u32 mask = 0xFFFFFFFFu;
          if (w == (start >> 5)) mask &= 0xFFFFFFFFu << (start & 31);
          if (w == (last >> 5)) mask &= 0xFFFFFFFFu >> (31 - (last & 31));
          atomicOr(reinterpret_cast<u32*>(lds_raw + bits_off + 4 * w), mask);
        }
        break;
      }
    }
  }
  if (bad) raise(c, ST_ROARING);
}

// Run lengths -> run starts (exclusive scan) -> one start flag per run (templated/RLE.hpp:164-191
// expands runs back to back; the start of run r is the sum of the lengths before it). Every
// thread owns a contiguous slice of runs, so one block-wide scan covers any run count.
// Set by bitmap_starts when it could not (or was not asked to) fill the rank table of its bitmap.
__device__ __forceinline__ u32 rank_pending_word(const Ctx& c) { return c.red_off + 52; }
constexpr u32 kRankFuseWords = 8;
__device__ __forceinline__ void bitmap_starts(const Ctx& c, const Step& s) {
  const Tab len = resolve_tab(c, uniform(s.a[0]));
  const u32 bits_off = seq_lds(c, uniform(s.a[1]));
  const u32 nbits = uniform(s.a[3]), runs = uniform(s.a[4]);
  const u32 per = (runs + kBlockThreads - 1) / kBlockThreads;
  const u32 r_lo = threadIdx.x * per < runs ? threadIdx.x * per : runs;
  const u32 r_hi = r_lo + per < runs ? r_lo + per : runs;
  u32 sum = 0, bad = 0;
  for (u32 r = r_lo; r < r_hi; r++) {
    const int l = (int)tab_ld32(len, r);
    bad |= l <= 0;
    sum += l > 0 ? (u32)l : 0u;
  }
  // The rank table (start flags in front of every 64-row word) falls out of the same scan: runs
  // 0..r start in front of word w exactly when start(r) < 64 w <= start(r + 1), so the thread that
  // owns run r writes r + 1 into the words its run reaches. A run that reaches many words would
  // serialise that: it is left to the rank step behind this one, which otherwise finds nothing to do.
  const u32 rank_seq = uniform(s.a[2]);
  const u32 rank_off = rank_seq == NONE ? 0u : seq_lds(c, rank_seq);
  const u32 words = nbits / 64 + 1;
  if (threadIdx.x == 0) {
    lds_st32(rank_pending_word(c), rank_seq == NONE ? 1u : 0u);
    if (rank_seq != NONE) lds_st32(rank_off, 0);
  }
  u32 total;
  u32 start = block_exclusive_scan(c, sum, &total);
  bool left_over = false;
  for (u32 r = r_lo; r < r_hi; r++) {
    const int l = (int)tab_ld32(len, r);
    if (start < nbits)
      bitmap_or(bits_off, start);
    else
      bad = 1;
    const u32 next = start + (l > 0 ? (u32)l : 0u);
    if (rank_seq != NONE) {
      const u32 w_lo = (start >> 6) + 1, w_hi = (next >> 6) < words - 1 ? (next >> 6) : words - 1;
      if (w_hi >= w_lo && w_hi - w_lo >= kRankFuseWords) {
        left_over = true;
      } else {
        for (u32 w = w_lo; w <= w_hi; w++) lds_st32(rank_off + 4 * w, r + 1);
      }
    }
    start = next;
  }
  if (left_over) lds_st32(rank_pending_word(c), 1);
  if (total != nbits && threadIdx.x == 0) bad = 1;
  if (bad) raise(c, ST_RLE_COUNTS);
}

// ---- step: SIMDFastPFor<8> + VariableByte stream -> sequence (integer/PBP.cpp:37-49, 92-109) --------
// One value of a SIMD pack of 128 at width b: value i lives in lane i % 4 at bit (i / 4) * b of
// that lane's stream; word k of lane L is word 4k + L of the pack.
__device__ __forceinline__ u32 simd_pack_get(__amdgpu_buffer_rsrc_t rsrc, u32 word0, u32 b, u32 i) {
  const u32 bit = (i >> 2) * b;
  const u32 w = word0 + 4 * (bit >> 5) + (i & 3);
  const u32 lo = __builtin_amdgcn_raw_buffer_load_b32(rsrc, w * 4, 0, 0);
  const u32 hi = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (w + 4) * 4, 0, 0);
  const u32 x = __builtin_amdgcn_alignbit(hi, lo, bit);
  return b >= 32 ? x : (x & ((1u << b) - 1u));
}

// VariableByte tail of up to 255 values (7 bits per byte, low group first, a set top bit ends a
// value): bytes are staged in LDS, every thread owns five of them, a block-wide scan of the
// terminator counts gives each value its index.
__device__ __forceinline__ void vbyte_tail_decode(const Ctx& c, __amdgpu_buffer_rsrc_t rsrc, u32 first_word, u32 W,
                                                  u32 stage_off, const Tab& out, u32 out_first, u32 expect) {
  const u32 tid = threadIdx.x;
  const u32 words = W > first_word ? (W - first_word < 320 ? W - first_word : 320) : 0;
  for (u32 w = tid; w < 320; w += kBlockThreads)
    lds_st32(stage_off + 4 * w, w < words ? __builtin_amdgcn_raw_buffer_load_b32(rsrc, (first_word + w) * 4, 0, 0) : 0u);
  __syncthreads();
  const u32 nbytes = words * 4;
  u32 cnt = 0;
  for (u32 j = 5 * tid; j < 5 * tid + 5 && j < nbytes; j++) cnt += lds_raw[stage_off + j] >> 7;
  u32 total;
  u32 idx = block_exclusive_scan(c, cnt, &total);
  for (u32 j = 5 * tid; j < 5 * tid + 5 && j < nbytes; j++) {
    if (!(lds_raw[stage_off + j] & 128)) continue;
    u32 start = j;  // first byte of the value that ends at j: behind the previous terminator
    while (start > 0 && j - start < 4 && !(lds_raw[stage_off + start - 1] & 128)) start--;
    u32 v = 0;
    for (u32 q = start; q <= j; q++) v += (u32)(lds_raw[stage_off + q] & 127) << (7 * (q - start));
    if (idx < expect) tab_st32(out, out_first + idx, v);
    idx++;
  }
  if (total != expect && tid == 0) raise(c, ST_BP_TAIL);
  __syncthreads();
}

// Wait of a wave that streams packed words into its LDS ring and stores unpacked values in
// between: everything but the newest `newer` vector memory instructions (loads and stores
// complete in issue order on one counter) has completed. The wait count is an immediate, so one
// case per count; counts beyond the counter's range wait for a little more than needed.
__device__ __forceinline__ void pfor_wait(u32 newer) {
#define BTR_PFOR_WAIT_CASE(k) \
  case k: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(k) : "memory"); break;
#define BTR_PFOR_WAIT_CASE8(k)                                                                          \
  BTR_PFOR_WAIT_CASE(k) BTR_PFOR_WAIT_CASE(k + 1) BTR_PFOR_WAIT_CASE(k + 2) BTR_PFOR_WAIT_CASE(k + 3) \
  BTR_PFOR_WAIT_CASE(k + 4) BTR_PFOR_WAIT_CASE(k + 5) BTR_PFOR_WAIT_CASE(k + 6) BTR_PFOR_WAIT_CASE(k + 7)
  switch (newer < 63 ? newer : 63) {
    BTR_PFOR_WAIT_CASE8(0)
    BTR_PFOR_WAIT_CASE8(8)
    BTR_PFOR_WAIT_CASE8(16)
    BTR_PFOR_WAIT_CASE8(24)
    BTR_PFOR_WAIT_CASE8(32)
    BTR_PFOR_WAIT_CASE8(40)
    BTR_PFOR_WAIT_CASE8(48)
    BTR_PFOR_WAIT_CASE8(56)
  }
#undef BTR_PFOR_WAIT_CASE8
#undef BTR_PFOR_WAIT_CASE
}

constexpr int kPforStorePolicy = kStoreNt;  // non-temporal, as the other roots (cached stores: 5 % slower)
__device__ __forceinline__ void pfor_decode(const Ctx& c, const Step& s) {
  const u32 tid = threadIdx.x;
  const u8* base = c.src + uniform(s.a[0]);
  const u32 W = uniform(s.a[1]);
  const u32 desc0 = c.consts_off + 4 * uniform(s.a[2]);
  const u32 pages = uniform(s.a[3]) & PFOR_PAGES_MASK;
  const bool delta = (uniform(s.a[3]) & PFOR_DELTA_BIT) != 0;
  const bool may_merge = (uniform(s.a[3]) & PFOR_NO_MERGE) == 0;  // the work area holds the merge tables
  const Tab out = resolve_tab(c, uniform(s.a[4]));
  const u32 work = seq_lds(c, uniform(s.a[5]));
  const u32 table_off = work, bytes_lds = work + 4096, tail_off = work + 8192, packed_lds = work + 9472,
            ebase_off = tail_off,  // 257 words; the tail stage is only used after the pages
            jump_lds = packed_lds, pos_lds = packed_lds + 2 * (4096 + 512),  // dead before the rings fill
            exc_off = work + 25856, mask_off = exc_off + 4 * kPforExcCap, erank_off = mask_off + 32 * kPforSubBlocks;
  const u32 n = out.count;
  const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(base, W * 4);
  u32 bad = 0;
  for (u32 p = 0; p < pages; p++) {
    const u32 d = desc0 + 4 * kPforPageWords * p;
    const u32 first = uniform(lds32(d)), len = uniform(lds32(d + 4)), data_word = uniform(lds32(d + 8)),
              bytes_off = uniform(lds32(d + 12)), bytes_size = uniform(lds32(d + 16));
    const u32 blocks = len / btrfmt::PFOR_BLOCK;
    // 1. the byte container is a chain of variable-length records {b, exception count, [max bits,
    //    positions...]}. Staged in LDS (when it fits) the chain is resolved by pointer doubling:
    //    J0[o] = o + length of the record that would start at byte o; after round t thread r <
    //    2^(t+1) knows the offset of record r. Otherwise one lane follows it through global memory.
    //    Each block notes {b, count, exception width}, the offset of its position bytes and (below)
    //    the first word of its packs.
    const bool staged = bytes_size <= 4096;
    const u32 skew = bytes_off & 3;  // bytes_off is word aligned by construction; keep it general
    if (staged) {
      for (u32 w = tid; w * 4 < bytes_size + skew; w += kBlockThreads)
        lds_st32(bytes_lds + 4 * w, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (bytes_off - skew) + 4 * w, 0, 0));
      __syncthreads();
      constexpr u32 kDead = 4096;  // offsets from here on map to themselves
      auto jt = [&](u32 o) { return jump_lds + 2 * o; };
#pragma unroll
      for (u32 i = 0; i < 16; i++) {
        if (i * kBlockThreads >= bytes_size) break;  // offsets behind the container are never visited
        const u32 o = i * kBlockThreads + tid;
        u32 j = kDead + (o & 255);
        if (o + 2 <= bytes_size) {
          const u32 cex = lds_raw[bytes_lds + skew + o + 1];
          j = o + 2 + (cex ? 1 + cex : 0);
        }
        *reinterpret_cast<u16*>(lds_raw + jt(o)) = (u16)j;
      }
      *reinterpret_cast<u16*>(lds_raw + jt(kDead + tid)) = (u16)(kDead + tid);
      *reinterpret_cast<u16*>(lds_raw + jt(kDead + 256 + tid)) = (u16)(kDead + 256 + tid);
      if (tid == 0) *reinterpret_cast<u16*>(lds_raw + pos_lds) = 0;
      __syncthreads();
      // Four records per round: with the table at J^step, records step..4*step-1 follow from the ones
      // below step by one to three table steps, then the table goes to J^(4*step) (not after the last
      // round). Half the barriers of a round per doubling.
      for (u32 step = 1; step < blocks; step *= 4) {
        if (tid >= step && tid < 4 * step && tid < blocks) {
          const u32 q = tid / step;  // 1..3
          u32 v = lds16(jt(lds16(pos_lds + 2 * (tid - q * step))));
          if (q > 1) v = lds16(jt(v));
          if (q > 2) v = lds16(jt(v));
          *reinterpret_cast<u16*>(lds_raw + pos_lds + 2 * tid) = (u16)v;
        }
        if (4 * step >= blocks) break;  // every thread reads its own record offset below
        auto j4 = [&](u32 o) { return lds16(jt(lds16(jt(lds16(jt(lds16(jt(o)))))))); };
        u32 nj[8];  // J^(4*step) of this thread's sixteen offsets, two per register
#pragma unroll
        for (u32 i = 0; i < 8; i++)
          if (2 * i * kBlockThreads < bytes_size)
            nj[i] = j4(2 * i * kBlockThreads + tid) | (j4((2 * i + 1) * kBlockThreads + tid) << 16);
        __syncthreads();  // every thread has read the table (and the record offsets it needed)
#pragma unroll
        for (u32 i = 0; i < 8; i++)
          if (2 * i * kBlockThreads < bytes_size) {
            *reinterpret_cast<u16*>(lds_raw + jt(2 * i * kBlockThreads + tid)) = (u16)nj[i];
            *reinterpret_cast<u16*>(lds_raw + jt((2 * i + 1) * kBlockThreads + tid)) = (u16)(nj[i] >> 16);
          }
        __syncthreads();
      }
      // one thread per block reads its record; the packs' first words are a prefix sum of 8 * b
      u32 b = 0, cex = 0, k = 0, o = 0;
      if (tid < blocks) {
        o = lds16(pos_lds + 2 * tid);
        if (o + 2 <= bytes_size) {
          b = lds_raw[bytes_lds + skew + o];
          cex = lds_raw[bytes_lds + skew + o + 1];
          const u32 maxb = (cex && o + 3 <= bytes_size) ? lds_raw[bytes_lds + skew + o + 2] : 0u;
          if (cex) {
            if (maxb <= b || maxb > 32) bad = 1;
            k = maxb > b ? maxb - b : 1;
          }
          if (o + 2 + (cex ? 1 + cex : 0) > bytes_size) bad = 1;
        } else {
          bad = 1;
        }
        if (b > 32) {
          bad = 1;
          b = 32;
        }
      }
      u32 total;
      const u32 before = block_exclusive_scan(c, tid < blocks ? 8 * b : 0u, &total);
      if (tid < blocks) {
        lds_st32(table_off + 16 * tid, b | (cex << 8) | (k << 16));
        lds_st32(table_off + 16 * tid + 4, bytes_off + o + 2 + (cex ? 1 : 0));
        lds_st32(table_off + 16 * tid + 12, data_word + before);
      }
    } else if (tid == 0) {
      u32 o = 0, dw = data_word;
      for (u32 r = 0; r < blocks; r++) {
        u32 b = 0, cex = 0, k = 0, maxb = 0;
        if (o + 3 <= bytes_size) {
          b = base[bytes_off + o];
          cex = base[bytes_off + o + 1];
          maxb = base[bytes_off + o + 2];
        } else if (o + 2 <= bytes_size) {
          b = base[bytes_off + o];
          cex = base[bytes_off + o + 1];
        } else {
          bad = 1;
        }
        o += 2;
        if (cex) {
          if (maxb <= b || maxb > 32) bad = 1;
          k = maxb > b ? maxb - b : 1;
          o += 1;
        }
        if (b > 32) {
          bad = 1;
          b = 32;
        }
        lds_st32(table_off + 16 * r, b | (cex << 8) | (k << 16));
        lds_st32(table_off + 16 * r + 4, bytes_off + o);
        lds_st32(table_off + 16 * r + 12, dw);
        o += cex;
        dw += 8 * b;
      }
      if (o > bytes_size) bad = 1;
    }
    __syncthreads();
    BTR_STAMP(c, 9);
    // exceptions in front of each block (for step 4)
    u32 total_exc;
    {
      const u32 cex = tid < blocks ? (lds32(table_off + 16 * tid) >> 8) & 0xFF : 0u;
      const u32 before = block_exclusive_scan(c, cex, &total_exc);
      if (tid <= blocks) lds_st32(ebase_off + 4 * tid, before);
    }
    // 2. index of a block's first exception inside its width class: exceptions of the same width
    //    in the blocks before it
    //    A page rarely uses more than a few widths: one block-wide scan per width that occurs
    //    (a page has at most 256 blocks, one per thread); with many widths every block counts the
    //    blocks in front of it instead.
    u32 widths_used = 0;
    for (u32 k = 2; k <= 32; k++) widths_used += uniform(lds32(d + 4 * (5 + 31 + k - 2))) != 0;
    if (widths_used <= 8 && blocks <= kBlockThreads) {
      const u32 e = tid < blocks ? lds32(table_off + 16 * tid) : 0u;
      if (tid < blocks) lds_st32(table_off + 16 * tid + 8, 0);
      for (u32 k = 2; k <= 32; k++) {
        if (uniform(lds32(d + 4 * (5 + 31 + k - 2))) == 0) continue;
        const bool mine = tid < blocks && (e >> 16) == k;
        u32 total;
        const u32 before = block_exclusive_scan(c, mine ? (e >> 8) & 0xFF : 0u, &total);
        if (mine) lds_st32(table_off + 16 * tid + 8, before);
      }
    } else {
      for (u32 r = tid; r < blocks; r += kBlockThreads) {
        const u32 k = lds32(table_off + 16 * r) >> 16;
        u32 before = 0;
        for (u32 q = 0; q < r; q++) {
          const u32 e = lds32(table_off + 16 * q);
          before += (e >> 16) == k ? (e >> 8) & 0xFF : 0u;
        }
        lds_st32(table_off + 16 * r + 8, before);
      }
    }
    __syncthreads();  // the exception bases and class indexes are read by every thread from here on
    BTR_STAMP(c, 10);
    // 3. every wave takes a contiguous quarter of the blocks, whose packed words are contiguous
    //    too, and streams them through its own 4 KiB LDS ring in 1 KiB pieces (one direct-to-LDS
    //    load each, three pieces ahead): iteration p unpacks the blocks that start in piece p (they
    //    may run into piece p+1, which has landed); lane l extracts values l, l+64, l+128, l+192.
    //    Unpack and patches alternate over sub-ranges of the page's blocks: a 4-byte patch that
    //    arrives after its line has left the L2 costs a read-modify-write in memory.
    //    Exceptions are merged into the unpacked values before these are stored, so that every
    //    output line is written once: a 4-byte patch that arrives after its line has left the L2
    //    costs a read-modify-write in memory. Per sub-range of the page's blocks the exceptions'
    //    high bits go to LDS in list order and every block gets a 256-bit mask of patched positions
    //    (positions ascend inside a block, so the n-th set bit belongs to the block's n-th exception).
    //    A sub-range with more exceptions than the LDS list holds, and a table in LDS, are patched in
    //    place behind the unpack instead.
    constexpr u32 kSubBlocks = kPforSubBlocks;
    constexpr u32 kPatchUnroll = 4;  // exceptions per thread whose loads are in flight together
    constexpr u32 kNowhere = 0xFFFFFFF0u;
    const bool patch_global = out.loc != LOC_LDS;
    const __amdgpu_buffer_rsrc_t patch_rsrc = make_rsrc(patch_global ? out.g : nullptr, patch_global ? out.count * 4 : 0);
    // Where exception number en (of the page) lives: its block, position, and the two words that
    // hold its packed high bits. Dead lanes, exceptions without stored bits (width 1) and
    // out-of-range indexes point past the buffer descriptor (loads return 0, stores are dropped).
    struct ExcRef {
      u32 r, pos, lo_at, hi_at, sh, width;
      bool live, counted;
    };
    auto locate = [&](u32 en_raw, u32 e_last) {
      ExcRef x;
      x.counted = en_raw < e_last;
      bool live = x.counted;
      const u32 en = live ? en_raw : 0u;
      u32 r = 0;  // the last block whose first exception is at or before en
#pragma unroll
      for (u32 step = 128; step; step >>= 1) {
        const u32 cand = r + step;
        if (cand < blocks && lds32(ebase_off + 4 * (cand < blocks ? cand : 0u)) <= en) r = cand;
      }
      const u32 xi = en - lds32(ebase_off + 4 * r);
      const u32 e = lds32(table_off + 16 * r), pos_off = lds32(table_off + 16 * r + 4), ex0 = lds32(table_off + 16 * r + 8);
      const u32 b = e & 0xFF, k = e >> 16;
      live = live && b < 32;
      const u32 kc = k >= 2 ? k : 2u;
      const u32 cls_word = lds32(d + 4 * (5 + kc - 2));
      const u32 cls_size = k >= 2 ? lds32(d + 4 * (5 + 31 + kc - 2)) : 0u;
      const u32 full = cls_size & ~127u;
      x.pos = staged && pos_off + xi < bytes_off + bytes_size
                  ? (u32)lds_raw[bytes_lds + skew + (pos_off - bytes_off) + xi]
                  : __builtin_amdgcn_raw_buffer_load_b8(rsrc, pos_off + xi, 0, 0);  // past the payload: 0
      const u32 q = ex0 + xi;
      if (live && k >= 2 && q >= cls_size) bad = 1;
      const bool stored_bits = live && k >= 2 && q < cls_size;
      const bool simd = q < full;  // whole SIMD packs of 128 first, then plain packs of 32
      const u32 qq = q - full;
      const u32 bit = (simd ? (q & 127) >> 2 : qq & 31) * k;
      const u32 w = simd ? cls_word + (q >> 7) * 4 * k + 4 * (bit >> 5) + (q & 3)
                         : cls_word + (full >> 7) * 4 * k + (qq >> 5) * k + (bit >> 5);
      x.lo_at = stored_bits ? w * 4 : kNowhere;
      x.hi_at = stored_bits ? (w + (simd ? 4u : 1u)) * 4 : kNowhere;
      x.sh = (bit & 31) | (b << 8) | ((live && k < 2 ? 1u : 0u) << 16);
      x.width = stored_bits ? k : 0u;
      x.r = r;
      x.live = live;
      return x;
    };
    auto high_bits = [](const ExcRef& x, u32 lo, u32 hi) {
      const u32 a = __builtin_amdgcn_alignbit(hi, lo, x.sh & 31);
      const u32 k = x.width;
      const u32 v = (k >= 32 ? a : a & ((1u << k) - 1u)) | (x.sh >> 16);  // width 1: the bit is a one
      return v << ((x.sh >> 8) & 31);
    };
    for (u32 sb = 0; sb < blocks; sb += kSubBlocks) {
    const u32 se = sb + kSubBlocks < blocks ? sb + kSubBlocks : blocks;
    const u32 e_first = uniform(lds32(ebase_off + 4 * sb)), e_last = se < blocks ? uniform(lds32(ebase_off + 4 * se)) : total_exc;
    const bool merge = patch_global && may_merge && e_last - e_first <= kPforExcCap;
    if (merge) {
      for (u32 w = tid; w < 8 * kSubBlocks; w += kBlockThreads) lds_st32(mask_off + 4 * w, 0);
      __syncthreads();
      for (u32 e0 = e_first; e0 < e_last; e0 += kBlockThreads * kPatchUnroll) {
        ExcRef x[kPatchUnroll];
        u32 lo[kPatchUnroll], hi[kPatchUnroll];
#pragma unroll
        for (u32 u = 0; u < kPatchUnroll; u++) x[u] = locate(e0 + u * kBlockThreads + tid, e_last);
#pragma unroll
        for (u32 u = 0; u < kPatchUnroll; u++) {
          lo[u] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, x[u].lo_at, 0, 0);
          hi[u] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, x[u].hi_at, 0, 0);
        }
#pragma unroll
        for (u32 u = 0; u < kPatchUnroll; u++) {
          if (!x[u].counted) continue;
          // every listed exception gets its mask bit, so that ranks match list order
          lds_st32(exc_off + 4 * (e0 + u * kBlockThreads + tid - e_first), x[u].live ? high_bits(x[u], lo[u], hi[u]) : 0u);
          const u32 old = atomicOr(reinterpret_cast<u32*>(lds_raw + mask_off + 32 * (x[u].r - sb) + 4 * (x[u].pos >> 5)),
                                   1u << (x[u].pos & 31));
          if ((old >> (x[u].pos & 31)) & 1) bad = 1;  // two exceptions of a block at one position
        }
      }
      __syncthreads();
      if (tid < se - sb) {  // exceptions in front of each mask word, one byte each
        u32 before = 0, packed_lo = 0, packed_hi = 0;
#pragma unroll
        for (u32 j = 0; j < 8; j++) {
          (j < 4 ? packed_lo : packed_hi) |= (before & 0xFF) << (8 * (j & 3));
          before += (u32)__builtin_popcount(lds32(mask_off + 32 * tid + 4 * j));
        }
        lds_st64(erank_off + 8 * tid, (u64)packed_lo | ((u64)packed_hi << 32));
      }
      __syncthreads();
    }
    {
      const u32 lane = tid & 63, wave = uniform(tid >> 6);
      constexpr u32 kWaves = kBlockThreads / 64, kPiece = 256, kRing = 4 * kPiece;
      const u32 ring = packed_lds + 4 * kRing * wave;
      const u32 r_begin = sb + (se - sb) * wave / kWaves, r_end = sb + (se - sb) * (wave + 1) / kWaves;
      if (r_begin < r_end) {
        const u32 w_lo = uniform(lds32(table_off + 16 * r_begin + 12));
        const u32 last_e = uniform(lds32(table_off + 16 * (r_end - 1)));
        const u32 w_hi = uniform(lds32(table_off + 16 * (r_end - 1) + 12)) + 8 * (last_e & 0xFF) + 4;
        const u32 pieces = (w_hi - w_lo + kPiece - 1) / kPiece;
        auto issue = [&](u32 piece) {
          if (piece < pieces)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(lds_raw + ring + 4 * kPiece * (piece & 3)), 16,
                (w_lo + kPiece * piece + 4 * lane) * 4, 0, 0, 0);
        };
        issue(0);
        issue(1);
        issue(2);
        u32 r = r_begin;
        const bool to_global = out.loc != LOC_LDS;
        const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc(to_global ? out.g : nullptr, to_global ? out.count * 4 : 0);
        const bool wide = to_global && (reinterpret_cast<u64>(out.g) & 15) == 0;  // 16-byte stores
        u32 stored = 0, stored_before = 0;  // store instructions of the previous iteration and of the one before
        for (u32 piece = 0; piece < pieces && r < r_end; piece++) {
          // The slot of piece - 1: a block is at most one piece long, so the blocks that started in
          // it ended inside this one, and their values have been stored.
          issue(piece + 3);
          // Pieces <= piece + 1 must have landed. Behind the load of piece + 1 this wave issued the
          // stores of the last two iterations and the loads of the next two pieces (if any), and
          // exactly those may stay in flight.
          pfor_wait(stored + stored_before + (piece + 2 < pieces) + (piece + 3 < pieces));
          stored_before = stored;
          stored = 0;
          while (r < r_end) {
            const u32 rel = uniform(lds32(table_off + 16 * r + 12)) - w_lo;
            if (rel >= kPiece * (piece + 1)) break;
            const u32 b = uniform(lds32(table_off + 16 * r)) & 0xFF;
            const u32 mask = b >= 32 ? ~0u : (1u << b) - 1u;
            // lane l takes values 4l..4l+3 of the block: they sit at the same bit of four adjacent
            // words (one per 32-bit lane of the SIMD pack), so two 16-byte LDS reads fetch all of them
            const u32 bit = (lane & 31) * b;
            const u32 w = rel + (lane >> 5) * 4 * b + 4 * (bit >> 5);
            const u32x4 lo = *reinterpret_cast<const u32x4*>(lds_raw + ring + 4 * (w & (kRing - 1)));
            const u32x4 hi = *reinterpret_cast<const u32x4*>(lds_raw + ring + 4 * ((w + 4) & (kRing - 1)));
            u32x4 v;
            v.x = __builtin_amdgcn_alignbit(hi.x, lo.x, bit) & mask;
            v.y = __builtin_amdgcn_alignbit(hi.y, lo.y, bit) & mask;
            v.z = __builtin_amdgcn_alignbit(hi.z, lo.z, bit) & mask;
            v.w = __builtin_amdgcn_alignbit(hi.w, lo.w, bit) & mask;
            if (merge) {
              const u32 rb = r - sb;
              const u32 mword = lds32(mask_off + 32 * rb + 4 * (lane >> 3));
              const u32 nib = (mword >> (4 * (lane & 7))) & 15;
              if (nib) {
                u32 idx = uniform(lds32(ebase_off + 4 * r)) - e_first + lds_raw[erank_off + 8 * rb + (lane >> 3)] +
                          (u32)__builtin_popcount(mword & ((1u << (4 * (lane & 7))) - 1u));
                auto take = [&]() {
                  const u32 h = lds32(exc_off + 4 * (idx < kPforExcCap ? idx : kPforExcCap - 1));
                  idx++;
                  return h;
                };
                if (nib & 1) v.x |= take();
                if (nib & 2) v.y |= take();
                if (nib & 4) v.z |= take();
                if (nib & 8) v.w |= take();
              }
            }
            const u32 o = (first + r * btrfmt::PFOR_BLOCK + 4 * lane) * 4;
            if (to_global) {
              if (wide) {
                __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, o, 0, kPforStorePolicy);
                stored++;
              } else {
                __builtin_amdgcn_raw_buffer_store_b32(v.x, out_rsrc, o, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(v.y, out_rsrc, o + 4, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(v.z, out_rsrc, o + 8, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(v.w, out_rsrc, o + 12, 0, 0);
                stored += 4;
              }
            } else {
              lds_st32(out.lds + o, v.x);
              lds_st32(out.lds + o + 4, v.y);
              lds_st32(out.lds + o + 8, v.z);
              lds_st32(out.lds + o + 12, v.w);
            }
            r++;
          }
        }
      }
    }
    __syncthreads();  // also orders the stores above before the patches below
    BTR_STAMP(c, 11);
    // 4. exceptions that were not merged, one thread each: exception e of the page belongs to the
    //    block r with ebase[r] <= e < ebase[r + 1]; (value >> b) << b is ORed back in at its position.
    //    Into global memory: no branch between a thread's loads, so the packed exception values and
    //    the low bits of all its exceptions are in flight together.
    for (u32 e0 = e_first; patch_global && !merge && e0 < e_last; e0 += kBlockThreads * kPatchUnroll) {
      ExcRef x[kPatchUnroll];
      u32 at[kPatchUnroll], lo[kPatchUnroll], hi[kPatchUnroll], old[kPatchUnroll];
#pragma unroll
      for (u32 u = 0; u < kPatchUnroll; u++) {
        x[u] = locate(e0 + u * kBlockThreads + tid, e_last);
        at[u] = x[u].live ? (first + x[u].r * btrfmt::PFOR_BLOCK + x[u].pos) * 4 : kNowhere;
      }
#pragma unroll
      for (u32 u = 0; u < kPatchUnroll; u++) {
        lo[u] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, x[u].lo_at, 0, 0);
        hi[u] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, x[u].hi_at, 0, 0);
        old[u] = __builtin_amdgcn_raw_buffer_load_b32(patch_rsrc, at[u], 0, 0);
      }
#pragma unroll
      for (u32 u = 0; u < kPatchUnroll; u++)
        __builtin_amdgcn_raw_buffer_store_b32(old[u] | high_bits(x[u], lo[u], hi[u]), patch_rsrc, at[u], 0, 0);
    }
    for (u32 e0 = e_first; !patch_global && e0 < e_last; e0 += kBlockThreads * kPatchUnroll) {
      u32 at[kPatchUnroll], hi_bits[kPatchUnroll], old[kPatchUnroll];
      bool live[kPatchUnroll];
#pragma unroll
      for (u32 u = 0; u < kPatchUnroll; u++) {
        const u32 en = e0 + u * kBlockThreads + tid;
        live[u] = en < e_last;
        at[u] = hi_bits[u] = old[u] = 0;
        if (!live[u]) continue;
        u32 r = 0, hi = blocks;
        while (hi - r > 1) {
          const u32 mid = (r + hi) >> 1;
          if (lds32(ebase_off + 4 * mid) <= en)
            r = mid;
          else
            hi = mid;
        }
        const u32 x = en - lds32(ebase_off + 4 * r);
        const u32 e = lds32(table_off + 16 * r), pos_off = lds32(table_off + 16 * r + 4), ex0 = lds32(table_off + 16 * r + 8);
        const u32 b = e & 0xFF, k = e >> 16;
        if (b >= 32) {
          live[u] = false;
          continue;
        }
        const u32 cls_word = k >= 2 ? lds32(d + 4 * (5 + k - 2)) : 0u;
        const u32 cls_size = k >= 2 ? lds32(d + 4 * (5 + 31 + k - 2)) : 0u;
        const u32 full = cls_size & ~127u;
        const u32 pos = staged && pos_off + x < bytes_off + bytes_size
                            ? (u32)lds_raw[bytes_lds + skew + (pos_off - bytes_off) + x]
                            : __builtin_amdgcn_raw_buffer_load_b8(rsrc, pos_off + x, 0, 0);  // past the payload: 0
        u32 v = 1;
        if (k >= 2) {
          const u32 q = ex0 + x;
          if (q >= cls_size) {
            bad = 1;
            v = 0;
          } else if (q < full) {
            v = simd_pack_get(rsrc, cls_word + (q >> 7) * 4 * k, k, q & 127);
          } else {
            const u32 qq = q - full;
            v = bp_extract(rsrc, cls_word + (full >> 7) * 4 * k + (qq >> 5) * k, k, qq & 31);
          }
        }
        at[u] = first + r * btrfmt::PFOR_BLOCK + pos;
        hi_bits[u] = v << b;
        old[u] = tab_ld32(out, at[u]);
      }
#pragma unroll
      for (u32 u = 0; u < kPatchUnroll; u++)
        if (live[u]) tab_st32(out, at[u], old[u] | hi_bits[u]);
    }
    __syncthreads();
    }
  }
  if (bad) raise(c, ST_BP_HEADER);
  const u32 rounded = n / btrfmt::PFOR_BLOCK * btrfmt::PFOR_BLOCK;
  if (n > rounded) {
    const u32 tail_word = uniform(lds32(desc0 + 4 * kPforPageWords * pages));
    vbyte_tail_decode(c, rsrc, tail_word, W, tail_off, out, rounded, n - rounded);
  }
  if (delta && n > 0) {
    // PFOR_DELTA: inclusive wrapping prefix sum, 256-value tiles with a running carry
    if (out.loc != LOC_LDS) {
      __threadfence_block();
      __syncthreads();
    }
    u32 carry = 0;
    for (u32 t0 = 0; t0 < n; t0 += kBlockThreads) {
      const u32 i = t0 + tid;
      const u32 v = i < n ? tab_ld32(out, i) : 0u;
      u32 total;
      const u32 ex = block_exclusive_scan(c, v, &total);
      if (i < n) tab_st32(out, i, carry + ex + v);
      carry += total;
    }
  }
}

// bit i = (pipe[i] == value); each wave writes the 64-bit ballot of its 64 consecutive rows.
// A bare bit-packed pipe is scanned a 128-value group per wave, two groups (four loads) in flight.
template <u32 THREADS = kBlockThreads>
__device__ __forceinline__ void bitmap_eq(const Ctx& c, const Step& s) {
  const IntEval e = resolve_int(c, uniform(s.a[0]));
  const Seq& bm = seq_at(c, uniform(s.a[1]));
  const u32 bits_off = c.table_base + uniform(bm.off), words = uniform(bm.count);
  const u32 nbits = uniform(s.a[3]), value = uniform(s.a[4]);
  const u32 lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  if (e.base == B_BP && e.n_post == 0) {
    const UnpackLane u = unpack_lane();
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(e.bp.base, e.bp.bytes);
    const u32 groups = (nbits + btrfmt::FBP_GROUP - 1) / btrfmt::FBP_GROUP;
    for (u32 G0 = 2 * wave; G0 < groups; G0 += 2 * (THREADS / 64)) {
      u32 x[2][2];
#pragma unroll
      for (u32 j = 0; j < 2; j++) {
        const u32 G = G0 + j < groups ? G0 + j : G0;
        const u64 en = lds64(e.bp.index_off + 8 * G);
        x[j][0] = global_extract(rsrc, (u32)en + 1, (u32)(en >> 32), u.sh_lo, u.pm_lo, u.k);
        x[j][1] = global_extract(rsrc, (u32)en + 1, (u32)(en >> 32), u.sh_hi, u.pm_hi, u.k);
      }
#pragma unroll
      for (u32 j = 0; j < 2; j++) {
#pragma unroll
        for (u32 h = 0; h < 2; h++) {
          const u32 i = (G0 + j) * btrfmt::FBP_GROUP + 64 * h + lane;
          const u32 v = i >= e.bp.rounded ? lds32(e.bp.tail_off + 4 * ((i - e.bp.rounded) & 31)) : x[j][h];
          const u64 m = __ballot(i < nbits && v == value);
          if (lane == 0 && G0 + j < groups && (i >> 6) < words) lds_st64(bits_off + 8 * (i >> 6), m);
        }
      }
    }
    return;
  }
  if (e.base == B_RLE && e.n_post == 0 && e.tab.count > 0) {
    // a word of the result per thread: the runs that start inside the word (bits of the run-start
    // bitmap) cut it into stretches, and a stretch is all ones when its run's value matches
    const u32 last = e.tab.count - 1;
    const u32 need = (nbits + 63) / 64 < words ? (nbits + 63) / 64 : words;
    for (u32 w = threadIdx.x; w < need; w += THREADS) {
      u64 rest = lds64(e.rk.bits_off + 8 * w);
      const u32 n = lds32(e.rk.rank_off + 4 * w);  // runs that start before this word
      u32 r = n ? n - 1 : 0, k = 0, pos = 0;
      u64 out = 0;
      for (;;) {
        const bool match = tab_ld32(e.tab, r < last ? r : last) == value;
        const u32 nxt = rest ? (u32)__builtin_ctzll(rest) : 64u;
        if (match && nxt > pos) out |= (nxt == 64 ? ~0ull : (1ull << nxt) - 1) & ~((1ull << pos) - 1);
        if (nxt == 64) break;
        r = n + k;  // the run that starts at nxt: rank_le - 1
        k++;
        pos = nxt;
        rest &= rest - 1;
      }
      const u32 left = nbits - 64 * w;
      if (left < 64) out &= (1ull << left) - 1;
      lds_st64(bits_off + 8 * w, out);
    }
    return;
  }
  u32 bad = 0;
  constexpr u32 kEqUnroll = 4;  // independent rows in flight per lane: the lookups are LDS-latency chains
  for (u32 base = 0; base < nbits; base += THREADS * kEqUnroll) {
    u32 v[kEqUnroll];
#pragma unroll
    for (u32 q = 0; q < kEqUnroll; q++) {
      const u32 i = base + q * THREADS + threadIdx.x;
      const u32 ii = i < nbits ? i : nbits - 1;
      v[q] = (e.base == B_RLE && e.n_post == 0) ? eval_int<B_RLE, true, PS_NONE>(e, ii, bad)
                                                : eval_int<GEN, true, PS_GEN>(e, ii, bad);
    }
#pragma unroll
    for (u32 q = 0; q < kEqUnroll; q++) {
      const u32 i = base + q * THREADS + threadIdx.x;
      const u64 m = __ballot(i < nbits && v[q] == value);
      if (lane == 0 && (i >> 6) <= (nbits >> 6) && (i >> 6) < words) lds_st64(bits_off + 8 * (i >> 6), m);
    }
  }
  report(c, bad);
}

// rank[w] = set bits in words [0, w): every thread owns a contiguous slice of words.
template <u32 THREADS = kBlockThreads>
__device__ __forceinline__ void bitmap_rank(const Ctx& c, const Step& s) {
  const Seq& bm = seq_at(c, uniform(s.a[1]));
  const u32 bits_off = c.table_base + uniform(bm.off), words = uniform(bm.count);
  const u32 rank_off = seq_lds(c, uniform(s.a[2]));
  // a bitmap of run starts: the start-flag step in front of this one has usually filled the table
  if (uniform(s.a[0]) == 1 && uniform(lds32(rank_pending_word(c))) == 0) return;
  const u32 per = (words + THREADS - 1) / THREADS;
  const u32 w_lo = threadIdx.x * per < words ? threadIdx.x * per : words;
  const u32 w_hi = w_lo + per < words ? w_lo + per : words;
  u32 sum = 0;
  for (u32 w = w_lo; w < w_hi; w++) sum += __popcll(lds64(bits_off + 8 * w));
  u32 total;
  u32 before = block_exclusive_scan<THREADS>(c, sum, &total);
  for (u32 w = w_lo; w < w_hi; w++) {
    lds_st32(rank_off + 4 * w, before);
    before += __popcll(lds64(bits_off + 8 * w));
  }
  const u32 expect = uniform(s.a[4]);
  if (expect != NONE && total != expect && threadIdx.x == 0) raise(c, uniform(s.a[5]));
}

// extern/RoaringBitmap.cpp:32-53: one byte per row, 1 = valid.
// WIDE: 16 rows per lane for Roaring maps too (the stream kernels; the generic kernels keep the
// short byte loop, whose registers every other program of theirs would pay for).
template <u32 THREADS = kBlockThreads, bool WIDE = false>
__device__ __forceinline__ void write_nullmap(const Ctx& c, const Step& s) {
  const u32 type = uniform(s.a[0]), n = uniform(s.a[3]);
  const u32 bits_off = uniform(s.a[1]) == NONE ? 0 : seq_lds(c, uniform(s.a[1]));
  if (type == btrfmt::NM_ALLONES || type == btrfmt::NM_ALLZEROS) {
    // constant map: 16 bytes per lane behind an aligned head
    const u32 fill = type == btrfmt::NM_ALLONES ? 0x01010101u : 0u;
    const u32 head = (u32)((16 - (reinterpret_cast<u64>(c.nullmap) & 15)) & 15);
    const u32 h = head < n ? head : n;
    const u32 body = (n - h) / 16;
    u32x4 v;
    v.x = v.y = v.z = v.w = fill;
    for (u32 q = threadIdx.x; q < body; q += THREADS)
      BTR_NT_STORE(v, reinterpret_cast<u32x4*>(c.nullmap + h + 16 * q));
    for (u32 i = threadIdx.x; i < h; i += THREADS) c.nullmap[i] = (u8)fill;
    for (u32 i = h + 16 * body + threadIdx.x; i < n; i += THREADS) c.nullmap[i] = (u8)fill;
    return;
  }
  const u32 flip = type == btrfmt::NM_REGULAR ? 0u : 0xFFFFu;
  u32 done = 0;
  if (WIDE && (reinterpret_cast<u64>(c.nullmap) & 15) == 0) {
    // sixteen rows per lane: four bits become four bytes by one multiply (bit k lands in byte k)
    const u32 body = n / 16;
    for (u32 q = threadIdx.x; q < body; q += THREADS) {
      const u32 bits = ((lds32(bits_off + 4 * (q >> 1)) >> (16 * (q & 1))) & 0xFFFF) ^ flip;
      u32x4 v;
      v.x = ((bits & 15) * 0x00204081u) & 0x01010101u;
      v.y = (((bits >> 4) & 15) * 0x00204081u) & 0x01010101u;
      v.z = (((bits >> 8) & 15) * 0x00204081u) & 0x01010101u;
      v.w = (((bits >> 12) & 15) * 0x00204081u) & 0x01010101u;
      BTR_NT_STORE(v, reinterpret_cast<u32x4*>(c.nullmap + 16 * q));
    }
    done = body * 16;
  }
  for (u32 i = done + threadIdx.x; i < n; i += THREADS) {
    const u32 bit = (u32)(lds64(bits_off + 8 * (i >> 6)) >> (i & 63)) & 1;
    c.nullmap[i] = (u8)(bit ^ (flip & 1));
  }
}

// The streamed root of a stream kernel's program; the steps behind it build and write the nullmap.
__device__ __forceinline__ u32 stream_root_index(const Ctx& c, const ProgramHeader& hdr) {
  u32 s = hdr.n_steps - 1;
  while (s > 0 && uniform(step_at(c, s).kind) != S_STREAM_ROOT) s--;
  return s;
}
// Nullmap steps behind the root: a constant fill, or the Roaring expansion (its bitmap and rank
// table live in the payload ring, which is idle by now).
template <u32 THREADS>
__device__ __forceinline__ void stream_trailing_nullmap(const Ctx& c, const ProgramHeader& hdr, u32 root_idx) {
  if (root_idx + 1 >= hdr.n_steps) return;
  __syncthreads();  // every wave is done with the ring
  for (u32 s = root_idx + 1; s < hdr.n_steps; s++) {
    const Step& st = step_at(c, s);
    switch (uniform(st.kind)) {
      case S_BITMAP_CLEAR: bitmap_clear<THREADS>(c, st); break;
      case S_BITMAP_SET: bitmap_set<THREADS>(c, st); break;
      case S_BITMAP_RANK: bitmap_rank<THREADS>(c, st); break;
      case S_NULLMAP: write_nullmap<THREADS, true>(c, st); break;
      default: break;
    }
    __syncthreads();
  }
}

// A launch entry is checked before the item index or the program place it names is used: an entry
// that is not one the host wrote (read at a wrong stride, from a stale or foreign buffer) would
// otherwise send the workgroup through arbitrary memory and loop bounds. Reported on the plan's
// first status word, the only one that can be named without trusting the entry.
__device__ __forceinline__ bool entry_ok(const LaunchEntry& e, u32* __restrict__ status) {
  const u32 item = uniform(e.item), off = uniform(e.prog_off), bytes = uniform(e.prog_bytes);
  const bool ok = uniform(e.check) == launch_entry_check(item, off, bytes) && bytes >= sizeof(ProgramHeader) &&
                  bytes <= kMaxProgramBytes && (bytes & 3) == 0 && (off & 3) == 0;
  if (!ok && threadIdx.x == 0) atomicOr(status, (u32)ST_PROGRAM);
  return ok;
}

// Copies the program into LDS at BASE and checks it against the launch entry and the work item
// before anything is interpreted: `ok` false means the workgroup must return at once (every loop
// bound, table offset and barrier count below comes out of the program).
template <u32 THREADS, u32 BASE>
__device__ __forceinline__ ProgramHeader load_program(const WorkItem& it, const LaunchEntry& entry,
                                                      const u8* __restrict__ programs, u32* __restrict__ status, Ctx& c,
                                                      bool& ok) {
  const u32 item = entry.item;
  const u32 tid = threadIdx.x;
  c.side = uniform(it.flags) & ITEM_SIDE_LAUNCH;
  base_priority(c);  // raised for a few chunks beside a big launch
  {
    // placed by the launch entry, not by the work item: the copy does not wait for the item's load
    const u32* prog = reinterpret_cast<const u32*>(programs + entry.prog_off);
    for (u32 o = tid; o < entry.prog_bytes / 4; o += THREADS) lds_st32(BASE + 4 * o, prog[o]);
    // The item's status word starts clean (later raise() calls are ordered behind the barrier). ST_PROGRAM
    // is left standing: on the plan's first word it may come from another workgroup of this launch whose
    // entry was bad; the host clears it when it reads the words.
    if (tid == 0) atomicAnd(status + item, (u32)ST_PROGRAM);
  }
  __syncthreads();
  ProgramHeader hdr = *reinterpret_cast<const ProgramHeader*>(lds_raw + BASE);
  hdr.n_steps = uniform(hdr.n_steps);
  hdr.n_bp_steps = uniform(hdr.n_bp_steps);
  hdr.n_pipes = uniform(hdr.n_pipes);
  hdr.n_seqs = uniform(hdr.n_seqs);
  hdr.table_base = uniform(hdr.table_base);
  hdr.consts_off = uniform(hdr.consts_off);
  {
    const u32 bytes = uniform(entry.prog_bytes);
    constexpr u32 kMany = 4096;  // tables of a program: far fewer in practice, and small enough for 32-bit products
    const u32 fixed = (u32)sizeof(ProgramHeader) + hdr.n_steps * (u32)sizeof(Step) + hdr.n_pipes * (u32)sizeof(Pipe) +
                      hdr.n_seqs * (u32)sizeof(Seq);
    ok = uniform(hdr.magic) == kProgramMagic && uniform(hdr.prog_bytes) == bytes && uniform(it.prog_bytes) == bytes &&
         uniform(it.prog_off) == uniform(entry.prog_off) && hdr.n_steps >= 1 && hdr.n_steps <= kMany &&
         hdr.n_bp_steps <= hdr.n_steps && hdr.n_pipes <= kMany && hdr.n_seqs <= kMany && hdr.consts_off == fixed &&
         fixed <= bytes && hdr.table_base >= BASE + ((bytes + 15) & ~15u) + 64 && hdr.table_base <= kLdsCapBytes;
    if (!ok && tid == 0) atomicOr(status + item, (u32)ST_PROGRAM);
  }
  c.src = reinterpret_cast<const u8*>(it.src);
  c.dst = reinterpret_cast<u8*>(it.dst);
  c.nullmap = reinterpret_cast<u8*>(it.nullmap_dst);
  c.scratch = reinterpret_cast<u8*>(it.scratch);
  c.status = status + item;
  c.steps_off = BASE + sizeof(ProgramHeader);
  c.pipes_off = c.steps_off + hdr.n_steps * sizeof(Step);
  c.seqs_off = c.pipes_off + hdr.n_pipes * sizeof(Pipe);
  c.table_base = hdr.table_base;
  c.red_off = hdr.table_base - 64;
  c.consts_off = BASE + hdr.consts_off;
  c.debug = it.debug_flags;
  c.debug_buf = it.debug_buf;
  return hdr;
}

// bytes (a multiple of 4) from global memory (any alignment) to LDS (16-byte aligned) by the
// THREADS threads of a stream kernel: four 16-byte pieces in flight per thread.
template <u32 THREADS>
__device__ __forceinline__ void copy_to_lds(u32 lds_off, const u8* __restrict__ g, u32 bytes) {
  const u32 pieces = bytes / 16;
  for (u32 i0 = 0; i0 < pieces; i0 += 4 * THREADS) {
    u32x4 v[4];
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
      const u32 i = i0 + k * THREADS + threadIdx.x;
      if (i < pieces) __builtin_memcpy(&v[k], g + 16ull * i, 16);
    }
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
      const u32 i = i0 + k * THREADS + threadIdx.x;
      if (i < pieces) *reinterpret_cast<u32x4*>(lds_raw + lds_off + 16 * i) = v[k];
    }
  }
  for (u32 o = pieces * 16 + 4 * threadIdx.x; o < bytes; o += 4 * THREADS) lds_st32(lds_off + o, ldg32(g + o));
}

// Lean kernel body for chunks whose program is [optional dictionary copy into LDS] + streamed
// bit-packed root: DICT(BP), FOR(BP) and plain BP integer columns.
template <u32 PS, bool NM>
__device__ __forceinline__ void stream_block(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                                             const u8* __restrict__ programs, u32* __restrict__ status) {
  const LaunchEntry entry = order[blockIdx.x];
  if (!entry_ok(entry, status)) return;
  const u32 item = entry.item;
  const WorkItem it = items[item];
  Ctx c;
  bool ok;
  const ProgramHeader hdr = load_program<kStreamThreads, kStreamProgBase>(it, entry, programs, status, c, ok);
  if (!ok) return;
  const u32 root_idx = NM ? stream_root_index(c, hdr) : hdr.n_steps - 1;
  const Step& root = step_at(c, root_idx);
  const u32 pipe_id = uniform(root.a[0]);
  const IntEval e = resolve_int(c, pipe_id);
  // the loading wave starts the first batches of the payload before the dictionary is copied: the two
  // memory latencies overlap (the ring lies in front of everything else in LDS)
  if (uniform(threadIdx.x >> 6) == kStreamWaves - 1) ring_issue_first<false, false>(e.bp, pipe_at(c, pipe_id));
  for (u32 s = 0; s < root_idx; s++) {  // dictionary copies: a bare source sequence -> LDS
    const Step& st = step_at(c, s);
    const Tab from = resolve_tab(c, uniform(pipe_at(c, uniform(st.a[0])).a[0]));
    const Tab to = resolve_tab(c, uniform(st.a[1]));
    copy_to_lds<kStreamThreads>(to.lds, from.g, to.count * to.esize);
  }
  __syncthreads();
  stream_int_bp<PS>(c, pipe_at(c, pipe_id), e, reinterpret_cast<u32*>(c.dst), jump_scratch(c, root));
  if (NM) stream_trailing_nullmap<kStreamThreads>(c, hdr, root_idx);
}

// DICT(BP) double chunks: [dictionary copy into LDS] + streamed root.
template <bool NM>
__device__ __forceinline__ void stream_dictd_block(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                                                   const u8* __restrict__ programs, u32* __restrict__ status) {
  const LaunchEntry entry = order[blockIdx.x];
  if (!entry_ok(entry, status)) return;
  const u32 item = entry.item;
  const WorkItem it = items[item];
  Ctx c;
  stamp(it, 0);
  bool ok;
  const ProgramHeader hdr = load_program<kDictdThreads, kStreamProgBase>(it, entry, programs, status, c, ok);
  if (!ok) return;
  const u32 root_idx = NM ? stream_root_index(c, hdr) : hdr.n_steps - 1;
  const Step& root = step_at(c, root_idx);
  if (uniform(threadIdx.x >> 6) == kDictdWaves - 1) {  // first payload batches beside the dictionary copy
    const Pipe& codes = pipe_at(c, uniform(pipe_at(c, uniform(root.a[0])).a[0]));
    ring_issue_first<false, false>(resolve_bp(c, codes), codes);
  }
  for (u32 s = 0; s < root_idx; s++) {
    const Step& st = step_at(c, s);
    const Tab from = resolve_tab(c, uniform(pipe_at(c, uniform(st.a[0])).a[0]));
    const Tab to = resolve_tab(c, uniform(st.a[1]));
    copy_to_lds<kDictdThreads>(to.lds, from.g, to.count * to.esize);
  }
  __syncthreads();
  stamp(it, 1);
  stream_dictd(c, pipe_at(c, uniform(root.a[0])), reinterpret_cast<u64*>(c.dst), jump_scratch(c, root));
  if (NM) stream_trailing_nullmap<kDictdThreads>(c, hdr, root_idx);
  stamp(it, 2);
}

// PSEUDODECIMAL(BP, BP, raw patches) chunks. The program is [clear bitmap, rank, streamed root]:
// the exception bitmap is filled by a ring pass over the exponent stream between the first two.
template <bool NM>
__device__ __forceinline__ void stream_pseudo_block(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                                                    const u8* __restrict__ programs, u32* __restrict__ status) {
  const LaunchEntry entry = order[blockIdx.x];
  if (!entry_ok(entry, status)) return;
  const u32 item = entry.item;
  const WorkItem it = items[item];
  Ctx c;
  stamp(it, 0);
  bool ok;
  const ProgramHeader hdr = load_program<kStreamThreads, kStreamProgBase>(it, entry, programs, status, c, ok);
  if (!ok) return;
  stamp(it, 1);
  const u32 root_idx = NM ? stream_root_index(c, hdr) : hdr.n_steps - 1;
  const Step& root = step_at(c, root_idx);
  const Pipe& rp = pipe_at(c, uniform(root.a[0]));
  const Pipe& pd = pipe_at(c, uniform(rp.a[0]));
  const Pipe& pe = pipe_at(c, uniform(rp.a[1]));
  constexpr u32 frac_off = kStreamFracBase;
  const u32 jump_off = jump_scratch(c, root);
  RankView rk;
  rk.bits_off = seq_lds(c, uniform(rp.a[3]));
  rk.rank_off = seq_lds(c, uniform(rp.a[4]));
  const u32 n = uniform(rp.count);
  // the powers-of-ten table, indexed by stream value & 31: by exponent (entries above the largest
  // exponent repeat it) or, for dictionary-coded exponents, by code
  ExpCode xc{btrfmt::DEC_EXCEPTION_CODE, 5, 0};
  {
    const u32 l = threadIdx.x & 63;
    u32 ev = l;  // exponent behind stream value l
    if (uniform(pe.n_post) != 0) {
      const Tab dict = resolve_tab(c, uniform(pe.post[0].arg));
      xc.codes = dict.count;
      ev = l < dict.count ? tab_ld32(dict, l) : btrfmt::DEC_MAX_EXPONENT;
      const u64 hit = __builtin_amdgcn_ballot_w64(l < dict.count && ev == btrfmt::DEC_EXCEPTION_CODE);
      xc.exception = hit ? (u32)__builtin_ctzll(hit) : NONE;
      xc.need_bits = !hit ? 33u : xc.exception ? 32 - (u32)__builtin_clz(xc.exception) : 0u;
    }
    if (threadIdx.x < 32)
      lds_st64(frac_off + 8 * threadIdx.x,
               (u64)__double_as_longlong(kFractions[ev < btrfmt::DEC_MAX_EXPONENT ? ev : btrfmt::DEC_MAX_EXPONENT]));
  }
  // The patches are gathered row by row during the digit pass: their lines are pulled into L2 now, one
  // load per 128-byte line with the value unused, so that the gathers do not go to HBM one by one.
  u32 warm = 0;
  {
    const Tab pt = resolve_tab(c, uniform(rp.a[2]));
    const u32 lines = (pt.count * 8 + 127) / 128;
    if (pt.loc != LOC_LDS && pt.count && lines <= kStreamThreads) {
      const u64 first = reinterpret_cast<u64>(pt.g) & ~3ull, last = (reinterpret_cast<u64>(pt.g) + pt.count * 8 - 4) & ~3ull;
      if (threadIdx.x < lines) {
        const u64 a = first + 128ull * threadIdx.x;
        warm = ldg32(reinterpret_cast<const u8*>(a < last ? a : last));
      }
    }
  }
  bitmap_clear<kStreamThreads>(c, step_at(c, root_idx - 2));
  __syncthreads();
  stamp(it, 2);
  stream_exception_bits(c, pe, resolve_bp(c, pe), rk, n / 64 + 1, n, jump_off, xc);
  asm volatile("" ::"v"(warm));  // keeps the loads above
  stamp(it, 3);
  // the exponent pass ended with a barrier and nothing reads the ring until the digit pass: its
  // loading wave starts the first digit batches now, so that they land while the ranks are built
  if (uniform(threadIdx.x >> 6) == kStreamWaves - 1) ring_issue_first<true, true>(resolve_bp(c, pd), pd);
  bitmap_rank<kStreamThreads>(c, step_at(c, root_idx - 1));
  __syncthreads();
  stamp(it, 4);
  stream_pseudo(c, rp, pd, pe, rk, reinterpret_cast<u64*>(c.dst), frac_off, jump_off, xc);
  if (NM) stream_trailing_nullmap<kStreamThreads>(c, hdr, root_idx);
  stamp(it, 5);
}

constexpr u32 kWarmLines = 2 * kBlockThreads;  // chunks of up to 64 KiB (bigger ones are streamed through anyway: warming them cost PFOR 3 %)
constexpr u32 kWarmFromLines = 64;             // up to 8 KiB the header windows of the first step cover the chunk anyway

template <bool DOUBLE_KERNEL, bool WITH_PFOR>
__device__ __forceinline__ void decode_block(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                                    const u8* __restrict__ programs, u32* __restrict__ status) {
  const LaunchEntry entry = order[blockIdx.x];
  if (!entry_ok(entry, status)) return;
  const u32 item = entry.item;
  const WorkItem it = items[item];
  const u32 tid = threadIdx.x;
  Ctx c;
  stamp(it, 0);
  // A small chunk is pulled into this XCD's L2 while the program is loaded: one load per 128-byte
  // line, the values unused. The preparation steps are chains of dependent loads from the chunk
  // (header windows, header words, run values, run lengths ...), and each of them then takes an L2
  // round trip instead of one to HBM.
  u32 warm = 0;
  {
    const u32 lines = (uniform(it.src_bytes) + 127) / 128;
    if (lines > kWarmFromLines && lines <= kWarmLines) {
      // whole words inside the chunk's bytes (a chunk is at least its 12-byte header; allocations start aligned)
      const u64 first = it.src & ~3ull, last = (it.src + it.src_bytes - 4) & ~3ull;
      for (u32 l = tid; l < lines; l += kBlockThreads) {
        const u64 a = first + 128ull * l;
        warm ^= ldg32(reinterpret_cast<const u8*>(a < last ? a : last));
      }
    }
  }
  bool ok;
  const ProgramHeader hdr = load_program<kBlockThreads, 0>(it, entry, programs, status, c, ok);
  if (!ok) return;
  stamp(it, 1);

  // bit-unpack indexes: one stream per wave, streams in parallel
#ifdef BTR_DEBUG_STEPS
  if (!(it.debug_flags & 5))
#endif
    for (u32 s = uniform(tid >> 6); s < hdr.n_bp_steps; s += kBlockThreads / 64)
      bp_index_wave(c, step_at(c, s));
  __syncthreads();
  stamp(it, 2);

  for (u32 s = hdr.n_bp_steps; s < hdr.n_steps; s++) {
    const Step& st = step_at(c, s);
#ifdef BTR_DEBUG_STEPS  // step skipping leaves the column undecoded: diagnostic builds only
    if ((it.debug_flags & 2) && s + 1 == hdr.n_steps) break;
    if ((it.debug_flags & 4) && s + 1 != hdr.n_steps) continue;  // root step only, over whatever LDS holds
    if ((it.debug_flags >> 8) && s - hdr.n_bp_steps + 1 >= (it.debug_flags >> 8)) break;  // stop after k-1 steps
#endif
    switch (uniform(st.kind)) {
      case S_MATERIALIZE: materialize<DOUBLE_KERNEL>(c, uniform(st.a[0]), uniform(st.a[1])); break;
      case S_BITMAP_CLEAR: bitmap_clear(c, st); break;
      case S_BITMAP_SET: bitmap_set(c, st); break;
      case S_BITMAP_STARTS: bitmap_starts(c, st); break;
      case S_BITMAP_EQ: bitmap_eq(c, st); break;
      case S_BITMAP_RANK: bitmap_rank(c, st); break;
      case S_NULLMAP: write_nullmap(c, st); break;
      case S_PFOR_DECODE:
        if (WITH_PFOR) pfor_decode(c, st);
        break;
      default: break;
    }
    if (!uniform(st.no_barrier)) __syncthreads();
    stamp(it, 3 + s - hdr.n_bp_steps);
  }
  asm volatile("" ::"v"(warm));  // keeps the warm-up loads; nothing waits for them on their own account
}

}  // namespace

// `order` lists the work items this launch decodes (integer and double columns are launched
// separately); status words are indexed by work item.
extern "C" __global__ void __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(6, 8)))
btr_decode_int_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                      const u8* __restrict__ programs, u32* __restrict__ status) {
  decode_block<false, false>(items, order, programs, status);
}

// 128 VGPRs (four waves per SIMD): at 96 the kernel spilled 38 registers.
extern "C" __global__ void __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(4, 8)))
btr_decode_dbl_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                      const u8* __restrict__ programs, u32* __restrict__ status) {
  decode_block<true, false>(items, order, programs, status);
}

// Streamed bit-packed integer roots, one kernel per post-op signature (plan.h: KERNEL_STREAM_*).
extern "C" __global__ void __launch_bounds__(kStreamThreads)
btr_stream_none_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                       const u8* __restrict__ programs, u32* __restrict__ status) {
  stream_block<PS_NONE, false>(items, order, programs, status);
}
extern "C" __global__ void __launch_bounds__(kStreamThreads)
btr_stream_add_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                      const u8* __restrict__ programs, u32* __restrict__ status) {
  stream_block<PS_ADD, false>(items, order, programs, status);
}
extern "C" __global__ void __launch_bounds__(kStreamThreads)
btr_stream_dict_lds_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                           const u8* __restrict__ programs, u32* __restrict__ status) {
  stream_block<PS_DICT_LDS, false>(items, order, programs, status);
}
extern "C" __global__ void __launch_bounds__(kStreamThreads)
btr_stream_dict_glb_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                           const u8* __restrict__ programs, u32* __restrict__ status) {
  stream_block<PS_DICT_GLB, false>(items, order, programs, status);
}
// PSEUDODECIMAL(BP, BP, raw patches) double columns (plan.h: KERNEL_STREAM_PSEUDO).
extern "C" __global__ void __launch_bounds__(kStreamThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))
btr_stream_pseudo_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                         const u8* __restrict__ programs, u32* __restrict__ status) {
  stream_pseudo_block<false>(items, order, programs, status);
}
// DICT(BP) double columns (plan.h: KERNEL_STREAM_DICTD).
extern "C" __global__ void __launch_bounds__(kDictdThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))
btr_stream_dictd_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                        const u8* __restrict__ programs, u32* __restrict__ status) {
  stream_dictd_block<false>(items, order, programs, status);
}

// The generic kernels with the PFOR step (plan.h: KERNEL_PFOR_INT / KERNEL_PFOR_DBL).
// 128 VGPRs (four waves per SIMD): at 96 the integer kernel spilled 35 registers, and the spill
// stores were 14 % of the launch's write requests.
extern "C" __global__ void __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(4, 8)))
btr_decode_int_pfor_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                           const u8* __restrict__ programs, u32* __restrict__ status) {
  decode_block<false, true>(items, order, programs, status);
}

extern "C" __global__ void __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(4, 8)))
btr_decode_dbl_pfor_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                           const u8* __restrict__ programs, u32* __restrict__ status) {
  decode_block<true, true>(items, order, programs, status);
}

// The stream kernels with the nullmap steps behind the root (plan.h: stream kind + KERNEL_STREAM_NULLMAP).
#define BTR_STREAM_NULLMAP_KERNEL(name, threads, body)                                                     \
  extern "C" __global__ void __launch_bounds__(threads) __attribute__((amdgpu_waves_per_eu(8, 8)))         \
  name(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order, const u8* __restrict__ programs, \
       u32* __restrict__ status) {                                                                          \
    body(items, order, programs, status);                                                                   \
  }
BTR_STREAM_NULLMAP_KERNEL(btr_stream_none_nm_kernel, kStreamThreads, (stream_block<PS_NONE, true>))
BTR_STREAM_NULLMAP_KERNEL(btr_stream_add_nm_kernel, kStreamThreads, (stream_block<PS_ADD, true>))
BTR_STREAM_NULLMAP_KERNEL(btr_stream_dict_lds_nm_kernel, kStreamThreads, (stream_block<PS_DICT_LDS, true>))
BTR_STREAM_NULLMAP_KERNEL(btr_stream_dict_glb_nm_kernel, kStreamThreads, (stream_block<PS_DICT_GLB, true>))
BTR_STREAM_NULLMAP_KERNEL(btr_stream_pseudo_nm_kernel, kStreamThreads, (stream_pseudo_block<true>))
BTR_STREAM_NULLMAP_KERNEL(btr_stream_dictd_nm_kernel, kDictdThreads, (stream_dictd_block<true>))
#undef BTR_STREAM_NULLMAP_KERNEL

// One launch over chunks of several 512-thread stream kinds (a relation's FOR, DICT and
// Pseudodecimal columns; a column whose blocks took different cascades): every workgroup runs the
// block routine of its segment's kind. All of these kinds run four workgroups per CU whatever their
// LDS need, so sharing a launch (and its LDS size, the largest of them) costs none of them occupancy,
// while one launch instead of several leaves no gaps between kernels of one hardware queue.
namespace {
template <bool WITH_NULLMAP>
__device__ __forceinline__ void stream_mixed_block(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                                                   const u8* __restrict__ programs, u32* __restrict__ status,
                                                   const LaunchSegments& seg) {
  u32 kind = seg.kind[0];
#pragma unroll
  for (u32 k = 1; k < kMaxSegments; k++)
    if (k < seg.n && blockIdx.x >= seg.first_block[k]) kind = seg.kind[k];
  switch (uniform(kind)) {
    case KERNEL_STREAM_NONE: stream_block<PS_NONE, WITH_NULLMAP>(items, order, programs, status); break;
    case KERNEL_STREAM_ADD: stream_block<PS_ADD, WITH_NULLMAP>(items, order, programs, status); break;
    case KERNEL_STREAM_DICT_LDS: stream_block<PS_DICT_LDS, WITH_NULLMAP>(items, order, programs, status); break;
    case KERNEL_STREAM_DICT_GLB: stream_block<PS_DICT_GLB, WITH_NULLMAP>(items, order, programs, status); break;
    case KERNEL_STREAM_PSEUDO: stream_pseudo_block<WITH_NULLMAP>(items, order, programs, status); break;
    default:  // not a kind this kernel decodes: the launch was put together wrongly
      if (threadIdx.x == 0) atomicOr(status, (u32)ST_PROGRAM);
      break;
  }
}
}  // namespace
extern "C" __global__ void __launch_bounds__(kStreamThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))
btr_stream_mixed_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                        const u8* __restrict__ programs, u32* __restrict__ status, const LaunchSegments seg) {
  stream_mixed_block<false>(items, order, programs, status, seg);
}
extern "C" __global__ void __launch_bounds__(kStreamThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))
btr_stream_mixed_nm_kernel(const WorkItem* __restrict__ items, const LaunchEntry* __restrict__ order,
                           const u8* __restrict__ programs, u32* __restrict__ status, const LaunchSegments seg) {
  stream_mixed_block<true>(items, order, programs, status, seg);
}
