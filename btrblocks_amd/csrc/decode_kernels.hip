// HIP decode kernels for gfx950 (MI355X): one 256-thread workgroup per compressed column chunk,
// interpreting the flat program compiled by plan_builder.cpp (plan.h).
//
// What it replaces in the reference: the recursive virtual decompress() cascade
// (btrblocks/scheme/CompressionScheme.hpp:37-41, 70-74) with its scheme bodies
//   FastPFor FastBinaryPacking<32>+VariableByte decodeArray (via btrblocks/extern/FastPFOR.cpp:47-53)
//   templated/DynamicDictionary.hpp:108-216, templated/RLE.hpp:133-250,
//   templated/Frequency.hpp:99-136, templated/FixedDictionary.hpp:56-71,
//   integer/FOR.cpp:46-65, integer/OneValue.cpp:30-39, double/Pseudodecimal.cpp:348-452,
//   double/DoubleBP.cpp:41-57, extern/RoaringBitmap.cpp:32-114 (nullmap expansion).
//
// Shape of the computation (HBM-bound integer work, no MFMA):
//   1. the program is copied into LDS;
//   2. every bit-packed stream gets a group index {word offset, width header} in LDS: the
//      FastBinaryPacking layout interleaves one header word per 128 values with the payload, so
//      header positions form a chain; one wave per stream walks it through 256-word register
//      windows (coalesced, prefetched loads; wave-uniform readlane steps), different streams on
//      different waves;
//   3. short child streams (run lengths/values, exception lists, patches) are materialized into
//      LDS (or an HBM scratch slice), start-flag / exception bitmaps get a popcount rank table;
//   4. the root pipe is evaluated row by row - lane i of a wave produces row base+i, so stores
//      are contiguous - with full-length children (codes, deltas, digits, exponents) bit-unpacked
//      on the fly from the compressed stream; nothing but the decoded column is written to HBM.
// The evaluation loops are instantiated per base kind so that each loop only keeps the state
// of its own scheme in registers; integer and double columns run in separate kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "btr_format.h"
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
__device__ inline u32 ldg32(const u8* p) {  // any byte alignment
  u32 v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ inline u64 ldg64(const u8* p) {
  u64 v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
__device__ inline u32 ldg16(const u8* p) {
  u16 v;
  __builtin_memcpy(&v, p, 2);
  return v;
}
__device__ inline u32 lds32(u32 off) { return *reinterpret_cast<const u32*>(lds_raw + off); }
__device__ inline u64 lds64(u32 off) { return *reinterpret_cast<const u64*>(lds_raw + off); }
__device__ inline void lds_st32(u32 off, u32 v) { *reinterpret_cast<u32*>(lds_raw + off) = v; }
__device__ inline void lds_st64(u32 off, u64 v) { *reinterpret_cast<u64*>(lds_raw + off) = v; }
__device__ inline u32 uniform(u32 v) { return __builtin_amdgcn_readfirstlane(v); }

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
};

__device__ inline const Step& step_at(const Ctx& c, u32 i) {
  return *reinterpret_cast<const Step*>(lds_raw + c.steps_off + i * sizeof(Step));
}
__device__ inline const Pipe& pipe_at(const Ctx& c, u32 i) {
  return *reinterpret_cast<const Pipe*>(lds_raw + c.pipes_off + i * sizeof(Pipe));
}
__device__ inline const Seq& seq_at(const Ctx& c, u32 i) {
  return *reinterpret_cast<const Seq*>(lds_raw + c.seqs_off + i * sizeof(Seq));
}
__device__ inline void raise(const Ctx& c, u32 bits) { atomicOr(c.status, bits); }

// A resolved sequence: LDS byte offset or global pointer.
struct Tab {
  const u8* g;  // global base (scratch, source bytes or output); unused for LDS
  u32 lds;      // absolute LDS byte offset
  u32 count;
  u32 loc;
  u32 esize;
};

__device__ inline Tab resolve_tab(const Ctx& c, u32 seq_id) {
  Tab t;
  t.g = nullptr;
  t.lds = 0;
  t.count = 0;
  t.loc = LOC_LDS;
  t.esize = 4;
  if (seq_id == NONE) return t;
  const Seq s = seq_at(c, seq_id);
  t.count = s.count;
  t.loc = s.loc;
  t.esize = s.esize;
  switch (s.loc) {
    case LOC_LDS: t.lds = c.table_base + s.off; break;
    case LOC_SCRATCH: t.g = c.scratch + s.off; break;
    case LOC_SRC: t.g = c.src + s.off; break;
    case LOC_DST: t.g = c.dst + s.off; break;
    default: t.g = c.nullmap + s.off; break;
  }
  return t;
}
__device__ inline u32 tab_ld32(const Tab& t, u32 i) {
  if (t.loc == LOC_LDS) return lds32(t.lds + 4 * i);
  return ldg32(t.g + 4ull * i);
}
__device__ inline u64 tab_ld64(const Tab& t, u32 i) {
  if (t.loc == LOC_LDS) return lds64(t.lds + 8 * i);
  return ldg64(t.g + 8ull * i);
}
__device__ inline u32 tab_ld_small(const Tab& t, u32 i) {  // esize 1, 2 or 4
  if (t.esize == 4) return tab_ld32(t, i);
  if (t.loc == LOC_LDS)
    return t.esize == 1 ? lds_raw[t.lds + i] : *reinterpret_cast<const u16*>(lds_raw + t.lds + 2 * i);
  return t.esize == 1 ? (u32)t.g[i] : ldg16(t.g + 2ull * i);
}
__device__ inline void tab_st32(const Tab& t, u32 i, u32 v) {
  if (t.loc == LOC_LDS)
    lds_st32(t.lds + 4 * i, v);
  else
    *reinterpret_cast<u32*>(const_cast<u8*>(t.g) + 4ull * i) = v;  // scratch and output are aligned
}
__device__ inline void tab_st64(const Tab& t, u32 i, u64 v) {
  if (t.loc == LOC_LDS)
    lds_st64(t.lds + 8 * i, v);
  else
    *reinterpret_cast<u64*>(const_cast<u8*>(t.g) + 8ull * i) = v;
}

// ---- bitmap + rank ---------------------------------------------------------------------------
// bits: u64 words in LDS; rank[w] = set bits in words [0, w).
struct RankView {
  u32 bits_off, rank_off;
};
// set bits strictly below row i, and whether row i itself is set
__device__ inline u32 rank_lt(const RankView& r, u32 i, bool* member) {
  const u32 w = i >> 6, p = i & 63;
  const u64 word = lds64(r.bits_off + 8 * w);
  *member = (word >> p) & 1;
  return lds32(r.rank_off + 4 * w) + __popcll(word & ((1ull << p) - 1));
}
// set bits at or below row i
__device__ inline u32 rank_le(const RankView& r, u32 i) {
  const u32 w = i >> 6, p = i & 63;
  const u64 word = lds64(r.bits_off + 8 * w);
  return lds32(r.rank_off + 4 * w) + __popcll(word & ((2ull << p) - 1));
}

// ---- bit-packed stream view --------------------------------------------------------------------
struct BpView {
  const u32* words;  // payload (4-byte aligned), words[0] = packed value count
  u32 last_word;     // index of the last payload word (loads are clamped to it)
  u32 rounded;       // values packed by FastBinaryPacking (multiple of 32)
  u32 index_off;     // LDS offset of {word offset, header} per 128 values
  u32 tail_off;      // LDS offset of the decoded VariableByte tail
};

__device__ inline BpView resolve_bp(const Ctx& c, const Pipe& p) {
  BpView b;
  b.words = reinterpret_cast<const u32*>(c.src + p.a[0]);
  b.last_word = p.a[1] ? p.a[1] - 1 : 0;
  b.rounded = p.a[3];
  b.index_off = c.table_base + seq_at(c, p.a[2]).off;
  b.tail_off = p.a[4] == NONE ? 0 : c.table_base + seq_at(c, p.a[4]).off;
  return b;
}

// One value of a miniblock: `base_word` is the first word of the miniblock, value k occupies bits
// [k*b, (k+1)*b) of its little-endian bit stream.
__device__ inline u32 bp_extract(const BpView& e, u32 base_word, u32 b, u32 k) {
  const u32 bit = k * b;
  const u32 word = base_word + (bit >> 5);
  const u32 w0 = word < e.last_word ? word : e.last_word;
  const u32 w1 = word + 1 < e.last_word ? word + 1 : e.last_word;
  const u32 lo = e.words[w0], hi = e.words[w1];
  const u32 v = __builtin_amdgcn_alignbit(hi, lo, bit & 31);
  return b >= 32 ? v : (v & ((1u << b) - 1u));
}

// Value `i` of a FastBinaryPacking<32>+VariableByte stream. Group g = i/128 starts at word
// index[g].x with header index[g].y = (b0<<24)|(b1<<16)|(b2<<8)|b3; miniblock m = (i/32)%4 starts
// after the header and the b_j words of the miniblocks before it.
// ALIGNED: the wave's lanes evaluate 64 consecutive rows starting at a multiple of 64, so the
// group is wave-uniform and its index entry is fetched once.
template <bool ALIGNED>
__device__ inline u32 bp_get(const BpView& e, u32 i) {
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
  return bp_extract(e, off + 1 + before, b, k);
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

__device__ inline IntEval resolve_int(const Ctx& c, u32 pipe_id) {
  const Pipe& p = pipe_at(c, pipe_id);
  IntEval e;
  e.base = p.base;
  e.count = p.count;
  e.k0 = 0;
  e.bp = BpView{nullptr, 0, 0, 0, 0};
  e.rk = RankView{0, 0};
  e.tab = resolve_tab(c, NONE);
  switch (p.base) {
    case B_SEQ: e.tab = resolve_tab(c, p.a[0]); break;
    case B_ONE: e.k0 = p.a[0]; break;
    case B_BP: e.bp = resolve_bp(c, p); break;
    case B_RLE:
      e.rk.bits_off = c.table_base + seq_at(c, p.a[0]).off;
      e.rk.rank_off = c.table_base + seq_at(c, p.a[1]).off;
      e.tab = resolve_tab(c, p.a[2]);
      break;
    case B_FREQ:
      e.k0 = p.a[0];
      e.rk.bits_off = c.table_base + seq_at(c, p.a[2]).off;
      e.rk.rank_off = c.table_base + seq_at(c, p.a[3]).off;
      e.tab = resolve_tab(c, p.a[4]);
      break;
    default: break;
  }
  e.n_post = p.n_post;
#pragma unroll
  for (int k = 0; k < MAX_POST; k++) {
    e.post_type[k] = k < p.n_post ? p.post[k].type : 0;
    e.post_bias[k] = p.post[k].arg;
    e.post_tab[k] = resolve_tab(c, (k < p.n_post && p.post[k].type == POST_DICT) ? p.post[k].arg : NONE);
  }
  return e;
}

// Dictionary gathers and FOR bias adds queued behind the base value
// (templated/DynamicDictionary.hpp:125-158: dest[i] = dict[codes[i]]; integer/FOR.cpp:62-64).
__device__ inline u32 apply_posts(const Ctx& c, const IntEval& e, u32 v) {
#pragma unroll
  for (int k = 0; k < MAX_POST; k++) {
    if (k < (int)e.n_post) {
      if (e.post_type[k] == POST_DICT) {
        const Tab& d = e.post_tab[k];
        if (v >= d.count) {
          raise(c, ST_DICT_CODE);
          v = d.count ? d.count - 1 : 0;
        }
        v = d.count ? tab_ld32(d, v) : 0;
      } else {
        v += e.post_bias[k];
      }
    }
  }
  return v;
}

template <u32 BASE, bool ALIGNED>
__device__ inline u32 eval_int_base(const IntEval& e, u32 i) {
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

template <u32 BASE, bool ALIGNED>
__device__ inline u32 eval_int(const Ctx& c, const IntEval& e, u32 i) {
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
  return apply_posts(c, e, v);
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
};

__device__ inline DblEval resolve_dbl(const Ctx& c, u32 pipe_id) {
  const Pipe& p = pipe_at(c, pipe_id);
  DblEval e;
  e.base = p.base;
  e.count = p.count;
  e.k0 = e.k1 = 0;
  e.rk = RankView{0, 0};
  e.tab = resolve_tab(c, NONE);
  e.ia.base = e.ib.base = B_ONE;
  e.ia.n_post = e.ib.n_post = 0;
  switch (p.base) {
    case B_SEQ: e.tab = resolve_tab(c, p.a[0]); break;
    case B_ONE:
      e.k0 = p.a[0];
      e.k1 = p.a[1];
      break;
    case B_RLE:
      e.rk.bits_off = c.table_base + seq_at(c, p.a[0]).off;
      e.rk.rank_off = c.table_base + seq_at(c, p.a[1]).off;
      e.tab = resolve_tab(c, p.a[2]);
      break;
    case B_FREQ:
      e.k0 = p.a[0];
      e.k1 = p.a[1];
      e.rk.bits_off = c.table_base + seq_at(c, p.a[2]).off;
      e.rk.rank_off = c.table_base + seq_at(c, p.a[3]).off;
      e.tab = resolve_tab(c, p.a[4]);
      break;
    case B_DICTD:
      e.ia = resolve_int(c, p.a[0]);
      e.tab = resolve_tab(c, p.a[1]);
      break;
    case B_PSEUDO:
      if (p.a[0] != NONE) e.ia = resolve_int(c, p.a[0]);
      e.ib = resolve_int(c, p.a[1]);
      e.tab = resolve_tab(c, p.a[2]);
      e.rk.bits_off = c.table_base + seq_at(c, p.a[3]).off;
      e.rk.rank_off = c.table_base + seq_at(c, p.a[4]).off;
      e.k0 = p.a[5];
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
template <u32 BASE, u32 IA, u32 IB>
__device__ inline u64 eval_dbl(const Ctx& c, const DblEval& e, u32 i) {
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
    u32 code = eval_int<IA, true>(c, e.ia, i);
    if (code >= e.tab.count) {
      raise(c, ST_DICT_CODE);
      code = e.tab.count ? e.tab.count - 1 : 0;
    }
    return e.tab.count ? tab_ld64(e.tab, code) : 0;
  }
  if (BASE == B_PSEUDO) {
    // double/Pseudodecimal.cpp:424-436: exponent 23 takes the next patch, anything else is
    // double(int32 digits) * 10^-exponent - one exact conversion and one IEEE multiply.
    const u32 ex = eval_int<IB, true>(c, e.ib, i);
    bool member;
    const u32 xr = rank_lt(e.rk, i, &member);
    u32 di = i - xr;
    di = di < e.k0 ? di : (e.k0 ? e.k0 - 1 : 0);
    const int digits = e.k0 ? (int)eval_int<IA, false>(c, e.ia, di) : 0;
    u32 idx = ex & btrfmt::DEC_INDEX_MASK;
    if (ex != btrfmt::DEC_EXCEPTION_CODE && idx > btrfmt::DEC_MAX_EXPONENT) {
      raise(c, ST_EXPONENT);
    }
    idx = idx > btrfmt::DEC_MAX_EXPONENT ? btrfmt::DEC_MAX_EXPONENT : idx;
    u64 r = (u64)__double_as_longlong((double)digits * kFractions[idx]);
    if (ex == btrfmt::DEC_EXCEPTION_CODE) {
      const u32 pi = xr < e.tab.count ? xr : e.tab.count - 1;
      r = e.tab.count ? tab_ld64(e.tab, pi) : 0;
    }
    return r;
  }
  if (BASE == B_DBP) {
    const u32 lo = bp_get<false>(e.ia.bp, 2 * i), hi = bp_get<false>(e.ia.bp, 2 * i + 1);
    return ((u64)hi << 32) | lo;
  }
  return 0;
}

// ---- block-wide exclusive scan of one value per thread ------------------------------------------
__device__ inline u32 block_exclusive_scan(const Ctx& c, u32 v, u32* total) {
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
  for (u32 w = 0; w < kBlockThreads / 64; w++) {
    const u32 s = lds32(c.red_off + 4 * w);
    before += w < wave ? s : 0;
    sum += s;
  }
  *total = sum;
  return before + incl - v;
}

// ---- step: bit-unpack group index ---------------------------------------------------------------
// One wave per stream. A window is 256 payload words, four consecutive words per lane. Every
// word is turned into "group length if this word were a header" (1 + the sum of its four width
// bytes, one byte per word); the chain of real headers is then followed through the window with
// wave-uniform readlane steps. Windows are fetched four at a time, the next four while the
// current four are walked. Group start positions collect in a register (lane g%64 keeps group g)
// and are flushed to LDS every 64 groups; headers are gathered in parallel afterwards.
struct Window {
  u32 x0, x1, x2, x3;
};

__device__ inline Window load_window(const u32* words, u32 W, u32 wb, u32 lane) {
  Window w;
  const u32 w0 = wb + 4 * lane;
  if (w0 + 4 <= W) {
    uint4 v;
    __builtin_memcpy(&v, words + w0, 16);
    w.x0 = v.x, w.x1 = v.y, w.x2 = v.z, w.x3 = v.w;
  } else {
    w.x0 = w0 < W ? words[w0] : 0;
    w.x1 = w0 + 1 < W ? words[w0 + 1] : 0;
    w.x2 = w0 + 2 < W ? words[w0 + 2] : 0;
    w.x3 = w0 + 3 < W ? words[w0 + 3] : 0;
  }
  return w;
}

__device__ inline u32 group_lengths(const Window& w) {
  auto glen = [](u32 x) {
    const u32 s = 1 + __builtin_amdgcn_sad_u8(x, 0u, 0u);
    return s > 255 ? 255u : s;
  };
  return glen(w.x0) | (glen(w.x1) << 8) | (glen(w.x2) << 16) | (glen(w.x3) << 24);
}

constexpr u32 kWalkBatch = 4;  // windows in flight per batch

__device__ void bp_index_wave(const Ctx& c, const Pipe& p) {
  const u32 lane = threadIdx.x & 63;
  const u32* words = reinterpret_cast<const u32*>(c.src + p.a[0]);
  const u32 W = p.a[1], rounded = p.a[3], groups = p.a[5];
  const u32 index_off = c.table_base + seq_at(c, p.a[2]).off;
  u32 pos = 1, g = 0, mine = 0;
  Window cur[kWalkBatch], nxt[kWalkBatch];
#pragma unroll
  for (u32 k = 0; k < kWalkBatch; k++) cur[k] = load_window(words, W, 256 * k, lane);
  for (u32 bb = 0; g < groups && bb < W; bb += 256 * kWalkBatch) {
    const bool more = bb + 256 * kWalkBatch < W;
    if (more) {
#pragma unroll
      for (u32 k = 0; k < kWalkBatch; k++) nxt[k] = load_window(words, W, bb + 256 * (kWalkBatch + k), lane);
    }
#pragma unroll
    for (u32 k = 0; k < kWalkBatch; k++) {
      const u32 wb = bb + 256 * k;
      const u32 d = group_lengths(cur[k]);
      while (pos < wb + 256 && g < groups) {
        const u32 rel = pos - wb;
        const u32 dv = __builtin_amdgcn_readlane(d, rel >> 2);
        mine = lane == (g & 63) ? pos : mine;
        pos += (dv >> (8 * (rel & 3))) & 0xFF;
        g++;
        if ((g & 63) == 0) lds_st32(index_off + 8 * (g - 64 + lane), mine);
      }
    }
    if (more) {
#pragma unroll
      for (u32 k = 0; k < kWalkBatch; k++) cur[k] = nxt[k];
    }
  }
  u32 bad = g < groups;  // the stream ended before all groups were found
  const u32 found = g;
  if ((found & 63) && lane < (found & 63)) lds_st32(index_off + 8 * ((found & ~63u) + lane), mine);
  if (pos > W) {
    bad = 1;
    pos = W;
  }
  // parallel pass: fetch and validate the header word of every group
  for (u32 gg = lane; gg < groups; gg += 64) {
    u32 gp = gg < found ? lds32(index_off + 8 * gg) : 0;
    gp = gp < W ? gp : (W ? W - 1 : 0);
    const u32 hv = W ? words[gp] : 0;
    // every width byte must be <= 32
    if ((hv & 0xC0C0C0C0u) || ((hv + 0x5F5F5F5Fu) & 0x80808080u)) bad = 1;
    lds_st64(index_off + 8 * gg, ((u64)hv << 32) | gp);
  }
  if (lane == 0) lds_st64(index_off + 8 * groups, (u64)pos);
  if (bad) raise(c, ST_BP_HEADER);

  // VariableByte tail (fewer than 32 values): 7 bits per byte, low group first, top bit ends a value.
  if (p.a[4] != NONE) {
    const u32 tail_off = c.table_base + seq_at(c, p.a[4]).off;
    const u32 stream_values = p.base == B_DBP ? 2 * p.count : p.count;
    const u32 expect = stream_values - rounded;
    const u32 tail_words = W - pos < 64 ? W - pos : 64;
    const u32 tw = lane < tail_words ? words[pos + lane] : 0;
    u32 v = 0, shift = 0, produced = 0;
    for (u32 bi = 0; bi < tail_words * 4; bi++) {
      const u32 wv = __builtin_amdgcn_readlane(tw, bi >> 2);
      const u32 ch = (wv >> ((bi & 3) * 8)) & 0xFF;
      v += (ch & 127) << (shift & 31);
      shift += 7;
      if (ch & 128) {
        if (produced < 32 && lane == 0) lds_st32(tail_off + 4 * produced, v);
        produced++;
        v = 0;
        shift = 0;
      }
    }
    if (produced != expect && lane == 0) raise(c, ST_BP_TAIL);
  }
}

// ---- step: materialize a pipe into a sequence ----------------------------------------------------
constexpr u32 kUnroll = 4;

template <u32 BASE>
__device__ void mat_int(const Ctx& c, const IntEval& e, const Tab& out, u32 count) {
  const u32 tid = threadIdx.x;
  for (u32 base = 0; base < count; base += kBlockThreads * kUnroll) {
    u32 v[kUnroll];
#pragma unroll
    for (u32 u = 0; u < kUnroll; u++) {
      const u32 i = base + u * kBlockThreads + tid;
      v[u] = eval_int<BASE, true>(c, e, i < count ? i : count - 1);
    }
#pragma unroll
    for (u32 u = 0; u < kUnroll; u++) {
      const u32 i = base + u * kBlockThreads + tid;
      if (i < count) tab_st32(out, i, v[u]);
    }
  }
}

// Root loop for a bit-packed base: every wave takes whole 128-value groups, lane l produces
// values l and l+64 of the group (miniblocks l/32 and 2+l/32), so the header and the miniblock
// offsets are wave-uniform and a group's 128 values are stored as two contiguous 256-byte rows.
constexpr u32 kGroupsPerIter = 2;

__device__ void mat_int_bp(const Ctx& c, const IntEval& e, const Tab& out, u32 count) {
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const BpView& bp = e.bp;
  const u32 groups = (bp.rounded + btrfmt::FBP_GROUP - 1) / btrfmt::FBP_GROUP;
  const u32 k = lane & 31;
  const bool upper = lane >= 32;
  for (u32 g0 = wave * kGroupsPerIter; g0 < groups; g0 += (kBlockThreads / 64) * kGroupsPerIter) {
    u32 v[kGroupsPerIter][2];
#pragma unroll
    for (u32 j = 0; j < kGroupsPerIter; j++) {
      const u32 g = g0 + j < groups ? g0 + j : groups - 1;
      const u64 ih = lds64(bp.index_off + 8 * g);
      const u32 off = uniform((u32)ih), hdr = uniform((u32)(ih >> 32));
      u32 b0 = hdr >> 24, b1 = (hdr >> 16) & 0xFF, b2 = (hdr >> 8) & 0xFF, b3 = hdr & 0xFF;
      b0 = b0 > 32 ? 32 : b0;
      b1 = b1 > 32 ? 32 : b1;
      b2 = b2 > 32 ? 32 : b2;
      b3 = b3 > 32 ? 32 : b3;
      const u32 first = off + 1;
      // miniblocks 0/1 (value lane) and 2/3 (value lane + 64)
      const u32 w_lo = upper ? first + b0 : first;
      const u32 w_hi = upper ? first + b0 + b1 + b2 : first + b0 + b1;
      v[j][0] = apply_posts(c, e, bp_extract(bp, w_lo, upper ? b1 : b0, k));
      v[j][1] = apply_posts(c, e, bp_extract(bp, w_hi, upper ? b3 : b2, k));
    }
#pragma unroll
    for (u32 j = 0; j < kGroupsPerIter; j++) {
      const u32 i = (g0 + j) * btrfmt::FBP_GROUP + lane;
      if (g0 + j < groups) {
        if (i < bp.rounded) tab_st32(out, i, v[j][0]);
        if (i + 64 < bp.rounded) tab_st32(out, i + 64, v[j][1]);
      }
    }
  }
  for (u32 i = bp.rounded + threadIdx.x; i < count; i += kBlockThreads)
    tab_st32(out, i, apply_posts(c, e, lds32(bp.tail_off + 4 * ((i - bp.rounded) & 31))));
}

template <u32 BASE, u32 IA, u32 IB>
__device__ void mat_dbl(const Ctx& c, const DblEval& e, const Tab& out, u32 count) {
  const u32 tid = threadIdx.x;
  for (u32 base = 0; base < count; base += kBlockThreads * kUnroll) {
    u64 v[kUnroll];
#pragma unroll
    for (u32 u = 0; u < kUnroll; u++) {
      const u32 i = base + u * kBlockThreads + tid;
      v[u] = eval_dbl<BASE, IA, IB>(c, e, i < count ? i : count - 1);
    }
#pragma unroll
    for (u32 u = 0; u < kUnroll; u++) {
      const u32 i = base + u * kBlockThreads + tid;
      if (i < count) tab_st64(out, i, v[u]);
    }
  }
}

template <bool DOUBLE_KERNEL>
__device__ void materialize(const Ctx& c, u32 pipe_id, u32 seq_id) {
  const u32 count = pipe_at(c, pipe_id).count;
  const bool is_double = pipe_at(c, pipe_id).is_double;
  const Tab out = resolve_tab(c, seq_id);
  if (count == 0) return;
  if (!is_double) {
    const IntEval e = resolve_int(c, pipe_id);
    switch (e.base) {
      case B_SEQ: mat_int<B_SEQ>(c, e, out, count); break;
      case B_ONE: mat_int<B_ONE>(c, e, out, count); break;
      case B_BP: mat_int_bp(c, e, out, count); break;
      case B_RLE: mat_int<B_RLE>(c, e, out, count); break;
      case B_FREQ: mat_int<B_FREQ>(c, e, out, count); break;
      default: break;
    }
  } else if (DOUBLE_KERNEL) {
    const DblEval e = resolve_dbl(c, pipe_id);
    switch (e.base) {
      case B_SEQ: mat_dbl<B_SEQ, GEN, GEN>(c, e, out, count); break;
      case B_ONE: mat_dbl<B_ONE, GEN, GEN>(c, e, out, count); break;
      case B_RLE: mat_dbl<B_RLE, GEN, GEN>(c, e, out, count); break;
      case B_FREQ: mat_dbl<B_FREQ, GEN, GEN>(c, e, out, count); break;
      case B_DICTD:
        if (e.ia.base == B_BP)
          mat_dbl<B_DICTD, B_BP, GEN>(c, e, out, count);
        else
          mat_dbl<B_DICTD, GEN, GEN>(c, e, out, count);
        break;
      case B_PSEUDO:
        if (e.ib.base == B_BP && (e.ia.base == B_BP || e.k0 == 0))
          mat_dbl<B_PSEUDO, B_BP, B_BP>(c, e, out, count);
        else
          mat_dbl<B_PSEUDO, GEN, GEN>(c, e, out, count);
        break;
      case B_DBP: mat_dbl<B_DBP, B_BP, GEN>(c, e, out, count); break;
      default: break;
    }
  }
}

// ---- bitmap steps ----------------------------------------------------------------------------------
__device__ inline void bitmap_or(u32 bits_off, u32 row) {
  atomicOr(reinterpret_cast<u32*>(lds_raw + bits_off + 4 * (row >> 5)), 1u << (row & 31));
}

__device__ void bitmap_clear(const Ctx& c, const Step& s) {
  const Seq bm = seq_at(c, s.a[1]);
  const u32 off = c.table_base + bm.off;
  for (u32 w = threadIdx.x; w < bm.count * 2; w += kBlockThreads) lds_st32(off + 4 * w, 0);
}

// Members of one CRoaring container (or of the whole bitmap for the u32-array form) -> bits.
__device__ void bitmap_set(const Ctx& c, const Step& s) {
  const u8* src = c.src + s.a[0];
  const u32 bits_off = c.table_base + seq_at(c, s.a[1]).off;
  const u32 count = s.a[2], first = s.a[3], nbits = s.a[4], kind = s.a[5];
  u32 bad = 0;
  for (u32 k = threadIdx.x; k < count; k += kBlockThreads) {
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
// expands runs back to back; the start of run r is the sum of the lengths before it).
__device__ void bitmap_starts(const Ctx& c, const Step& s) {
  const Tab len = resolve_tab(c, s.a[0]);
  const u32 bits_off = c.table_base + seq_at(c, s.a[1]).off;
  const u32 nbits = s.a[3], runs = s.a[4];
  u32 carry = 0, bad = 0;
  for (u32 base = 0; base < runs; base += kBlockThreads) {
    const u32 r = base + threadIdx.x;
    const int l = r < runs ? (int)tab_ld32(len, r) : 0;
    if (r < runs && l <= 0) bad = 1;
    u32 total;
    const u32 start = carry + block_exclusive_scan(c, l > 0 ? (u32)l : 0u, &total);
    if (r < runs) {
      if (start < nbits)
        bitmap_or(bits_off, start);
      else
        bad = 1;
    }
    carry += total;
  }
  if (carry != nbits && threadIdx.x == 0) bad = 1;
  if (bad) raise(c, ST_RLE_COUNTS);
}

// bit i = (pipe[i] == value); each wave writes the 64-bit ballot of its 64 consecutive rows.
__device__ void bitmap_eq(const Ctx& c, const Step& s) {
  const IntEval e = resolve_int(c, s.a[0]);
  const u32 bits_off = c.table_base + seq_at(c, s.a[1]).off;
  const u32 nbits = s.a[3], value = s.a[4];
  for (u32 base = 0; base < nbits; base += kBlockThreads) {
    const u32 i = base + threadIdx.x;
    const u32 ii = i < nbits ? i : nbits - 1;
    const u32 v = e.base == B_BP ? eval_int<B_BP, true>(c, e, ii) : eval_int<GEN, true>(c, e, ii);
    const u64 m = __ballot(i < nbits && v == value);
    if ((threadIdx.x & 63) == 0 && (i >> 6) <= (nbits >> 6)) lds_st64(bits_off + 8 * (i >> 6), m);
  }
}

__device__ void bitmap_rank(const Ctx& c, const Step& s) {
  const Seq bm = seq_at(c, s.a[1]);
  const u32 bits_off = c.table_base + bm.off;
  const u32 rank_off = c.table_base + seq_at(c, s.a[2]).off;
  u32 carry = 0;
  for (u32 base = 0; base < bm.count; base += kBlockThreads) {
    const u32 w = base + threadIdx.x;
    const u32 pc = w < bm.count ? __popcll(lds64(bits_off + 8 * w)) : 0;
    u32 total;
    const u32 ex = carry + block_exclusive_scan(c, pc, &total);
    if (w < bm.count) lds_st32(rank_off + 4 * w, ex);
    carry += total;
  }
  if (s.a[4] != NONE && carry != s.a[4] && threadIdx.x == 0) raise(c, s.a[5]);
}

// extern/RoaringBitmap.cpp:32-53: one byte per row, 1 = valid.
__device__ void write_nullmap(const Ctx& c, const Step& s) {
  const u32 type = s.a[0], n = s.a[3];
  const u32 bits_off = s.a[1] == NONE ? 0 : c.table_base + seq_at(c, s.a[1]).off;
  for (u32 i = threadIdx.x; i < n; i += kBlockThreads) {
    u8 v;
    if (type == btrfmt::NM_ALLONES) {
      v = 1;
    } else if (type == btrfmt::NM_ALLZEROS) {
      v = 0;
    } else {
      const u32 bit = (u32)(lds64(bits_off + 8 * (i >> 6)) >> (i & 63)) & 1;
      v = (u8)(type == btrfmt::NM_REGULAR ? bit : bit ^ 1);
    }
    c.nullmap[i] = v;
  }
}

template <bool DOUBLE_KERNEL>
__device__ inline void decode_block(const WorkItem* __restrict__ items, const u32* __restrict__ order,
                                    const u8* __restrict__ programs, u32* __restrict__ status) {
  const u32 item = order[blockIdx.x];
  const WorkItem it = items[item];
  const u32 tid = threadIdx.x;
  {
    const u32* prog = reinterpret_cast<const u32*>(programs + it.prog_off);
    for (u32 o = tid; o < it.prog_bytes / 4; o += kBlockThreads) lds_st32(4 * o, prog[o]);
    if (tid == 0) status[item] = 0;  // later raise() calls are ordered behind the barrier
  }
  __syncthreads();
  const ProgramHeader hdr = *reinterpret_cast<const ProgramHeader*>(lds_raw);
  Ctx c;
  c.src = reinterpret_cast<const u8*>(it.src);
  c.dst = reinterpret_cast<u8*>(it.dst);
  c.nullmap = reinterpret_cast<u8*>(it.nullmap_dst);
  c.scratch = reinterpret_cast<u8*>(it.scratch);
  c.status = status + item;
  c.steps_off = sizeof(ProgramHeader);
  c.pipes_off = c.steps_off + hdr.n_steps * sizeof(Step);
  c.seqs_off = c.pipes_off + hdr.n_pipes * sizeof(Pipe);
  c.table_base = hdr.table_base;
  c.red_off = hdr.table_base - 64;

  // bit-unpack indexes: one stream per wave, streams in parallel
  for (u32 s = tid >> 6; s < hdr.n_bp_steps; s += kBlockThreads / 64)
    bp_index_wave(c, pipe_at(c, step_at(c, s).a[0]));
  __syncthreads();

  for (u32 s = hdr.n_bp_steps; s < hdr.n_steps; s++) {
    const Step st = step_at(c, s);
    switch (st.kind) {
      case S_MATERIALIZE: materialize<DOUBLE_KERNEL>(c, st.a[0], st.a[1]); break;
      case S_BITMAP_CLEAR: bitmap_clear(c, st); break;
      case S_BITMAP_SET: bitmap_set(c, st); break;
      case S_BITMAP_STARTS: bitmap_starts(c, st); break;
      case S_BITMAP_EQ: bitmap_eq(c, st); break;
      case S_BITMAP_RANK: bitmap_rank(c, st); break;
      case S_NULLMAP: write_nullmap(c, st); break;
      default: break;
    }
    __syncthreads();
  }
}

}  // namespace

// `order` lists the work items this launch decodes (integer and double columns are launched
// separately); status words are indexed by work item.
extern "C" __global__ void __launch_bounds__(kBlockThreads)
btr_decode_int_kernel(const WorkItem* __restrict__ items, const u32* __restrict__ order,
                      const u8* __restrict__ programs, u32* __restrict__ status) {
  decode_block<false>(items, order, programs, status);
}

extern "C" __global__ void __launch_bounds__(kBlockThreads)
btr_decode_dbl_kernel(const WorkItem* __restrict__ items, const u32* __restrict__ order,
                      const u8* __restrict__ programs, u32* __restrict__ status) {
  decode_block<true>(items, order, programs, status);
}
