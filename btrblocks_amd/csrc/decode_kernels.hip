// HIP decode kernel for gfx950 (MI355X): one 256-thread workgroup per compressed column chunk,
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
//      windows (coalesced loads, wave-uniform readlane steps), different streams on different waves;
//   3. short child streams (run lengths/values, exception lists, patches) are materialized into
//      LDS (or an HBM scratch slice), start-flag / exception bitmaps get a popcount rank table;
//   4. the root pipe is evaluated row by row - lane i of a wave produces row base+i, so stores
//      are contiguous - with full-length children (codes, deltas, digits, exponents) bit-unpacked
//      on the fly from the compressed stream; nothing but the decoded column is written to HBM.
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
__device__ inline u32 rank_lt(u32 bits_off, u32 rank_off, u32 i, bool* member) {
  const u32 w = i >> 6, p = i & 63;
  const u64 word = lds64(bits_off + 8 * w);
  *member = (word >> p) & 1;
  return lds32(rank_off + 4 * w) + __popcll(word & ((1ull << p) - 1));
}

// ---- resolved integer pipe -------------------------------------------------------------------
struct IntEval {
  u32 base;
  u32 count;
  u32 k0;             // B_ONE value / B_FREQ top value / B_BP values packed by FastBinaryPacking
  const u32* words;   // B_BP payload (4-byte aligned)
  u32 last_word;      // B_BP: index of the last payload word
  u32 index_off;      // B_BP: LDS offset of {word offset, header} per 128 values
  u32 tail_off;       // B_BP: LDS offset of the decoded VariableByte tail
  u32 bits_off, rank_off;  // B_RLE / B_FREQ
  Tab tab;            // B_SEQ source / B_RLE run values / B_FREQ exceptions
  u32 n_post;
  u32 post_type[MAX_POST];
  u32 post_bias[MAX_POST];
  Tab post_tab[MAX_POST];
};

__device__ inline void resolve_bp(const Ctx& c, const Pipe& p, IntEval& e) {
  e.words = reinterpret_cast<const u32*>(c.src + p.a[0]);
  e.last_word = p.a[1] ? p.a[1] - 1 : 0;
  e.k0 = p.a[3];
  e.index_off = c.table_base + seq_at(c, p.a[2]).off;
  e.tail_off = p.a[4] == NONE ? 0 : c.table_base + seq_at(c, p.a[4]).off;
}

__device__ inline IntEval resolve_int(const Ctx& c, u32 pipe_id) {
  const Pipe& p = pipe_at(c, pipe_id);
  IntEval e;
  e.base = p.base;
  e.count = p.count;
  e.k0 = 0;
  e.words = nullptr;
  e.last_word = 0;
  e.index_off = e.tail_off = e.bits_off = e.rank_off = 0;
  e.tab = resolve_tab(c, NONE);
  switch (p.base) {
    case B_SEQ: e.tab = resolve_tab(c, p.a[0]); break;
    case B_ONE: e.k0 = p.a[0]; break;
    case B_BP: resolve_bp(c, p, e); break;
    case B_RLE:
      e.bits_off = c.table_base + seq_at(c, p.a[0]).off;
      e.rank_off = c.table_base + seq_at(c, p.a[1]).off;
      e.tab = resolve_tab(c, p.a[2]);
      break;
    case B_FREQ:
      e.k0 = p.a[0];
      e.bits_off = c.table_base + seq_at(c, p.a[2]).off;
      e.rank_off = c.table_base + seq_at(c, p.a[3]).off;
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

// Value `i` of a FastBinaryPacking<32>+VariableByte stream. Group g = i/128 starts at word
// index[g].x with header index[g].y = (b0<<24)|(b1<<16)|(b2<<8)|b3; miniblock m = (i/32)%4 starts
// after the header and the b_j words of the miniblocks before it; value k = i%32 occupies bits
// [k*b, (k+1)*b) of the miniblock's little-endian bit stream.
__device__ inline u32 bp_get(const IntEval& e, u32 i) {
  if (i >= e.k0) return lds32(e.tail_off + 4 * ((i - e.k0) & 31));
  const u64 ih = lds64(e.index_off + 8 * (i >> 7));
  const u32 off = (u32)ih, hdr = (u32)(ih >> 32);
  const u32 m = (i >> 5) & 3, k = i & 31;
  u32 b = (hdr >> (24 - 8 * m)) & 0xFF;
  b = b > 32 ? 32 : b;
  const u32 before = __builtin_amdgcn_sad_u8((u32)((u64)hdr >> (32 - 8 * m)), 0u, 0u);
  const u32 bit = k * b;
  const u32 word = off + 1 + before + (bit >> 5);
  const u32 w0 = word < e.last_word ? word : e.last_word;
  const u32 w1 = word + 1 < e.last_word ? word + 1 : e.last_word;
  const u32 lo = e.words[w0], hi = e.words[w1];
  const u32 v = __builtin_amdgcn_alignbit(hi, lo, bit & 31);
  return b >= 32 ? v : (v & ((1u << b) - 1u));
}

__device__ inline u32 eval_int(const Ctx& c, const IntEval& e, u32 i) {
  u32 v;
  switch (e.base) {
    case B_SEQ: v = tab_ld_small(e.tab, i); break;
    case B_ONE: v = e.k0; break;
    case B_BP: v = bp_get(e, i); break;
    case B_RLE: {
      // run index = (run starts at or before row i) - 1
      const u32 w = i >> 6, p = i & 63;
      const u64 word = lds64(e.bits_off + 8 * w);
      u32 r = lds32(e.rank_off + 4 * w) + __popcll(word & ((2ull << p) - 1));
      r = r ? r - 1 : 0;
      r = r < e.tab.count ? r : e.tab.count - 1;
      v = tab_ld32(e.tab, r);
      break;
    }
    case B_FREQ: {
      bool member;
      u32 r = rank_lt(e.bits_off, e.rank_off, i, &member);
      v = e.k0;
      if (member) {
        r = r < e.tab.count ? r : e.tab.count - 1;
        v = tab_ld32(e.tab, r);
      }
      break;
    }
    default: v = 0; break;
  }
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

// ---- resolved double pipe --------------------------------------------------------------------
struct DblEval {
  u32 base;
  u32 count;
  u32 k0, k1;          // B_ONE / B_FREQ value bits; B_PSEUDO: k0 = converted_count
  u32 bits_off, rank_off;
  Tab tab;             // B_SEQ / B_RLE values / B_FREQ exceptions / B_DICTD dictionary / B_PSEUDO patches
  IntEval ia;          // B_DICTD codes / B_PSEUDO digits / B_DBP word stream
  IntEval ib;          // B_PSEUDO exponents
};

__device__ inline DblEval resolve_dbl(const Ctx& c, u32 pipe_id) {
  const Pipe& p = pipe_at(c, pipe_id);
  DblEval e;
  e.base = p.base;
  e.count = p.count;
  e.k0 = e.k1 = 0;
  e.bits_off = e.rank_off = 0;
  e.tab = resolve_tab(c, NONE);
  switch (p.base) {
    case B_SEQ: e.tab = resolve_tab(c, p.a[0]); break;
    case B_ONE:
      e.k0 = p.a[0];
      e.k1 = p.a[1];
      break;
    case B_RLE:
      e.bits_off = c.table_base + seq_at(c, p.a[0]).off;
      e.rank_off = c.table_base + seq_at(c, p.a[1]).off;
      e.tab = resolve_tab(c, p.a[2]);
      break;
    case B_FREQ:
      e.k0 = p.a[0];
      e.k1 = p.a[1];
      e.bits_off = c.table_base + seq_at(c, p.a[2]).off;
      e.rank_off = c.table_base + seq_at(c, p.a[3]).off;
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
      e.bits_off = c.table_base + seq_at(c, p.a[3]).off;
      e.rank_off = c.table_base + seq_at(c, p.a[4]).off;
      e.k0 = p.a[5];
      break;
    case B_DBP:
      e.ia.base = B_BP;
      e.ia.n_post = 0;
      resolve_bp(c, p, e.ia);
      break;
    default: break;
  }
  return e;
}

__device__ inline u64 eval_dbl(const Ctx& c, const DblEval& e, u32 i) {
  switch (e.base) {
    case B_SEQ: return tab_ld64(e.tab, i);
    case B_ONE: return ((u64)e.k1 << 32) | e.k0;
    case B_RLE: {
      const u32 w = i >> 6, p = i & 63;
      const u64 word = lds64(e.bits_off + 8 * w);
      u32 r = lds32(e.rank_off + 4 * w) + __popcll(word & ((2ull << p) - 1));
      r = r ? r - 1 : 0;
      r = r < e.tab.count ? r : e.tab.count - 1;
      return tab_ld64(e.tab, r);
    }
    case B_FREQ: {
      bool member;
      u32 r = rank_lt(e.bits_off, e.rank_off, i, &member);
      u64 v = ((u64)e.k1 << 32) | e.k0;
      if (member) {
        r = r < e.tab.count ? r : e.tab.count - 1;
        v = tab_ld64(e.tab, r);
      }
      return v;
    }
    case B_DICTD: {
      u32 code = eval_int(c, e.ia, i);
      if (code >= e.tab.count) {
        raise(c, ST_DICT_CODE);
        code = e.tab.count ? e.tab.count - 1 : 0;
      }
      return e.tab.count ? tab_ld64(e.tab, code) : 0;
    }
    case B_PSEUDO: {
      // double/Pseudodecimal.cpp:424-436: exponent 23 takes the next patch, anything else is
      // double(int32 digits) * 10^-exponent - one exact conversion and one IEEE multiply.
      const u32 ex = eval_int(c, e.ib, i);
      bool member;
      const u32 xr = rank_lt(e.bits_off, e.rank_off, i, &member);
      if (ex == btrfmt::DEC_EXCEPTION_CODE) {
        const u32 pi = xr < e.tab.count ? xr : e.tab.count - 1;
        return e.tab.count ? tab_ld64(e.tab, pi) : 0;
      }
      u32 di = i - xr;
      di = di < e.k0 ? di : (e.k0 ? e.k0 - 1 : 0);
      const int digits = e.k0 ? (int)eval_int(c, e.ia, di) : 0;
      u32 idx = ex & btrfmt::DEC_INDEX_MASK;
      if (idx > btrfmt::DEC_MAX_EXPONENT) {
        raise(c, ST_EXPONENT);
        idx = btrfmt::DEC_MAX_EXPONENT;
      }
      const double r = (double)digits * kFractions[idx];
      return (u64)__double_as_longlong(r);
    }
    case B_DBP: {
      const u32 lo = bp_get(e.ia, 2 * i), hi = bp_get(e.ia, 2 * i + 1);
      return ((u64)hi << 32) | lo;
    }
    default: return 0;
  }
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
// One wave per stream. Each iteration takes a 256-word window of the payload into registers
// (four consecutive words per lane), turns every word into "group length if this word were a
// header" (1 + sum of its four width bytes), and follows the chain of real headers through the
// window with wave-uniform readlane steps.
__device__ void bp_index_wave(const Ctx& c, const Pipe& p) {
  const u32 lane = threadIdx.x & 63;
  const u32* words = reinterpret_cast<const u32*>(c.src + p.a[0]);
  const u32 W = p.a[1], rounded = p.a[3], groups = p.a[5];
  const u32 index_off = c.table_base + seq_at(c, p.a[2]).off;
  u32 pos = 1, g = 0, bad = 0;
  u32 x0 = 0, x1 = 0, x2 = 0, x3 = 0;
  auto load_window = [&](u32 wb) {
    const u32 w0 = wb + 4 * lane;
    x0 = w0 < W ? words[w0] : 0;
    x1 = w0 + 1 < W ? words[w0 + 1] : 0;
    x2 = w0 + 2 < W ? words[w0 + 2] : 0;
    x3 = w0 + 3 < W ? words[w0 + 3] : 0;
  };
  for (u32 wb = 0; g < groups; wb += 256) {
    if (wb >= W) {
      bad = 1;
      break;
    }
    if (pos >= wb + 256) continue;
    load_window(wb);
    auto glen = [](u32 x) {
      const u32 s = 1 + __builtin_amdgcn_sad_u8(x, 0u, 0u);
      return s > 255 ? 255u : s;
    };
    const u32 d = glen(x0) | (glen(x1) << 8) | (glen(x2) << 16) | (glen(x3) << 24);
    while (pos < wb + 256 && g < groups) {
      const u32 rel = pos - wb;
      const u32 l = rel >> 2, k = rel & 3;
      const u32 dv = __builtin_amdgcn_readlane(d, l);
      const u32 xs = k == 0 ? x0 : (k == 1 ? x1 : (k == 2 ? x2 : x3));
      const u32 hv = __builtin_amdgcn_readlane(xs, l);
      if (lane == 0) lds_st64(index_off + 8 * g, ((u64)hv << 32) | pos);
      // every width byte must be <= 32
      if ((hv & 0xC0C0C0C0u) || ((hv + 0x5F5F5F5Fu) & 0x80808080u)) bad = 1;
      pos += (dv >> (8 * k)) & 0xFF;
      g++;
    }
  }
  if (pos > W) {
    bad = 1;
    pos = W;
  }
  if (lane == 0) lds_st64(index_off + 8 * groups, (u64)pos);
  if (bad && lane == 0) raise(c, ST_BP_HEADER);

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

__device__ void materialize(const Ctx& c, u32 pipe_id, u32 seq_id) {
  const u32 tid = threadIdx.x;
  const u32 count = pipe_at(c, pipe_id).count;
  const bool is_double = pipe_at(c, pipe_id).is_double;
  const Tab out = resolve_tab(c, seq_id);
  if (count == 0) return;
  if (!is_double) {
    const IntEval e = resolve_int(c, pipe_id);
    for (u32 base = 0; base < count; base += kBlockThreads * kUnroll) {
      u32 v[kUnroll];
#pragma unroll
      for (u32 u = 0; u < kUnroll; u++) {
        const u32 i = base + u * kBlockThreads + tid;
        v[u] = eval_int(c, e, i < count ? i : count - 1);
      }
#pragma unroll
      for (u32 u = 0; u < kUnroll; u++) {
        const u32 i = base + u * kBlockThreads + tid;
        if (i < count) tab_st32(out, i, v[u]);
      }
    }
  } else {
    const DblEval e = resolve_dbl(c, pipe_id);
    for (u32 base = 0; base < count; base += kBlockThreads * kUnroll) {
      u64 v[kUnroll];
#pragma unroll
      for (u32 u = 0; u < kUnroll; u++) {
        const u32 i = base + u * kBlockThreads + tid;
        v[u] = eval_dbl(c, e, i < count ? i : count - 1);
      }
#pragma unroll
      for (u32 u = 0; u < kUnroll; u++) {
        const u32 i = base + u * kBlockThreads + tid;
        if (i < count) tab_st64(out, i, v[u]);
      }
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

// Run lengths -> run starts (exclusive scan, in place) -> one start flag per run
// (templated/RLE.hpp:164-191 expands runs back to back; the start of run r is the sum of the
// lengths before it).
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
    const u32 v = eval_int(c, e, i < nbits ? i : nbits - 1);
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

}  // namespace

extern "C" __global__ void __launch_bounds__(kBlockThreads)
btr_decode_kernel(const WorkItem* __restrict__ items, const u8* __restrict__ programs,
                  u32* __restrict__ status) {
  const WorkItem it = items[blockIdx.x];
  const u32 tid = threadIdx.x;
  {
    const u32* prog = reinterpret_cast<const u32*>(programs + it.prog_off);
    for (u32 o = tid; o < it.prog_bytes / 4; o += kBlockThreads) lds_st32(4 * o, prog[o]);
    if (tid == 0) status[blockIdx.x] = 0;  // later raise() calls are ordered behind the barrier
  }
  __syncthreads();
  const ProgramHeader hdr = *reinterpret_cast<const ProgramHeader*>(lds_raw);
  Ctx c;
  c.src = reinterpret_cast<const u8*>(it.src);
  c.dst = reinterpret_cast<u8*>(it.dst);
  c.nullmap = reinterpret_cast<u8*>(it.nullmap_dst);
  c.scratch = reinterpret_cast<u8*>(it.scratch);
  c.status = status + blockIdx.x;
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
      case S_MATERIALIZE: materialize(c, st.a[0], st.a[1]); break;
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
