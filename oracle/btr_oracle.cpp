// TEST INFRASTRUCTURE ONLY. CPU restatement of the BtrBlocks decompression path, used as the
// checker for the HIP decode kernels and as the timed CPU baseline. Nothing in the shipped
// library may include, link or call this file; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg load the library built from it.
//
// The reference's own decoder cannot be compiled here: every translation unit on the path pulls
// in FastPFor, CRoaring and boost::dynamic_bitset headers that are fetched from the network at
// configure time. This file restates the algorithm scheme by scheme, scalar and recursive like
// the reference, each function citing the file:line it follows.
//
// PARITY UNPINNED for two third-party payload formats whose source is not in the reference tree
// and for which the reference's tests hold no golden bytes:
//   * FastPFor (lemire/FastPFor @ bc375056): CompositeCodec<FastBinaryPacking<32>,VariableByte>
//     and CompositeCodec<SIMDFastPFor<8>,VariableByte>
//   * CRoaring (RoaringBitmap/CRoaring @ b88b0024): roaring_bitmap_serialize (non-portable)
// Their layouts are restated from the libraries' published formats. Everything the reference
// itself defines (headers, offsets, cascade order, Pseudodecimal arithmetic) is pinned by the
// reference source and by the known-input vectors its tools hold (tests/golden/).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using s32 = int32_t;

struct DecodeError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// A read-only window over the compressed bytes; every access is range checked (the reference
// does no bounds checking, tools/examples/compression.cpp:75-77).
struct Span {
  const u8* p;
  size_t n;
  const u8* origin = nullptr;  // start of the chunk / block: a 16-byte boundary when it was compressed
  const u8* at(size_t off, size_t len) const {
    if (off > n || len > n - off) throw DecodeError("read past the end of the compressed data");
    return p + off;
  }
  Span from(size_t off) const {
    if (off > n) throw DecodeError("offset past the end of the compressed data");
    return Span{p + off, n - off, origin};
  }
  u8 b(size_t off) const { return *at(off, 1); }
  u16 h(size_t off) const {
    u16 v;
    memcpy(&v, at(off, 2), 2);
    return v;
  }
  u32 w(size_t off) const {
    u32 v;
    memcpy(&v, at(off, 4), 4);
    return v;
  }
  u64 q(size_t off) const {
    u64 v;
    memcpy(&v, at(off, 8), 8);
    return v;
  }
};

enum NullmapType : u8 { ALLONES = 0, ALLZEROS = 1, REGULAR = 2, FLIPPED = 3 };  // Units.hpp:45

// ------------------------------------------------------------------------------------------
// FastPFor payloads (third-party; see the header comment)
// ------------------------------------------------------------------------------------------

// fastunpack: 32 values of `b` bits, LSB-first across b consecutive words.
void unpack32(const Span& s, size_t word, u32 b, u32* out) {
  if (b == 0) {
    memset(out, 0, 32 * 4);
    return;
  }
  if (b > 32) throw DecodeError("bit width above 32");
  const u8* base = s.at(word * 4, (size_t)b * 4);
  u64 acc = 0;
  u32 have = 0, next = 0;
  const u64 mask = b == 32 ? 0xFFFFFFFFull : ((1ull << b) - 1);
  for (u32 i = 0; i < 32; i++) {
    if (have < b) {
      u32 wv;
      memcpy(&wv, base + 4 * next++, 4);
      acc |= (u64)wv << have;
      have += 32;
    }
    out[i] = (u32)(acc & mask);
    acc >>= b;
    have -= b;
  }
}

// VariableByte::decodeArray: 7 bits per byte, low group first, a set top bit ends the value;
// trailing bytes without a terminator (zero padding) yield nothing.
size_t vbyte_decode(const Span& s, size_t word_begin, size_t word_end, u32* out, size_t cap) {
  const u8* p = s.at(word_begin * 4, (word_end - word_begin) * 4);
  const u8* end = p + (word_end - word_begin) * 4;
  size_t produced = 0;
  u32 v = 0, shift = 0;
  for (; p < end; p++) {
    v += (u32)(*p & 127) << shift;
    shift += 7;
    if (*p & 128) {
      if (produced == cap) throw DecodeError("VariableByte tail longer than expected");
      out[produced++] = v;
      v = 0;
      shift = 0;
    }
  }
  return produced;
}

// CompositeCodec<FastBinaryPacking<32>, VariableByte>::decodeArray. `payload` starts at the
// length word; `words` is the stored u32 count. Decodes exactly `n` values.
void fbp_decode(const Span& payload, u32 words, u32* out, size_t n) {
  if (n == 0) return;
  if (words == 0) throw DecodeError("empty FastBinaryPacking payload");
  payload.at(0, (size_t)words * 4);
  const u32 rounded = payload.w(0);
  if (rounded > n || rounded % 32 != 0) throw DecodeError("bad FastBinaryPacking length word");
  size_t pos = 1;
  for (u32 done = 0; done < rounded;) {
    const u32 hdr = payload.w(pos * 4);
    pos++;
    const u32 minis = std::min<u32>(4, (rounded - done) / 32);
    for (u32 m = 0; m < minis; m++) {
      const u32 b = (hdr >> (24 - 8 * m)) & 0xFF;
      unpack32(payload, pos, b, out + done);
      pos += b;
      done += 32;
    }
  }
  if (pos > words) throw DecodeError("FastBinaryPacking payload overruns its word count");
  const size_t tail = vbyte_decode(payload, pos, words, out + rounded, n - rounded);
  if (rounded + tail != n) throw DecodeError("FastPFor stream holds the wrong number of values");
}

// SIMD pack of 128 values at width b (FastPFor simdbitpacking): value i lives in lane i % 4 at
// bit (i / 4) * b of that lane's stream; word k of lane L is word 4k + L of the pack.
u32 simd_get(const Span& s, size_t word0, u32 b, u32 i) {
  if (b == 0) return 0;
  const u32 lane = i & 3, bit = (i >> 2) * b, k = bit >> 5, sh = bit & 31;
  u64 v = (u64)s.w((word0 + 4 * (size_t)k + lane) * 4) >> sh;
  if (sh + b > 32) v |= (u64)s.w((word0 + 4 * (size_t)(k + 1) + lane) * 4) << (32 - sh);
  return (u32)(b == 32 ? v : (v & ((1ull << b) - 1)));
}

// CompositeCodec<SIMDFastPFor<8>, VariableByte>::decodeArray (restated from FastPFor's
// simdfastpfor.h): pages of at most 65536 values, blocks of 256 values packed at width b as two
// SIMD packs, exceptions (value >> b) grouped by their width max_bits - b and patched back by
// position. `misalign` = (address of the payload / 4) % 4 when it was compressed: the codec pads
// to 16-byte boundaries of the absolute address.
void pfor_decode(const Span& payload, u32 words, u32* out, size_t n, u32 misalign) {
  if (n == 0) return;
  if (words == 0) throw DecodeError("empty PFOR payload");
  payload.at(0, (size_t)words * 4);
  const u32 rounded = payload.w(0);
  if (rounded > n || rounded % 256 != 0 || n - rounded >= 256) throw DecodeError("bad PFOR length word");
  size_t in = 1;
  for (u32 done = 0; done < rounded;) {
    const u32 len = std::min<u32>(65536, rounded - done);
    const size_t header = in;
    const size_t meta = header + payload.w(header * 4);
    const u32 bytesize = payload.w(meta * 4);
    const u8* bytes = payload.at((meta + 1) * 4, bytesize);
    size_t ex = meta + 1 + (bytesize + 3) / 4;
    const u32 bitmap = payload.w(ex * 4);
    ex++;
    std::vector<u32> classes[33];
    for (u32 k = 2; k <= 32; k++) {
      if (!(bitmap >> (k - 1) & 1)) continue;
      const u32 size = payload.w(ex * 4);
      ex++;
      classes[k].resize(size);
      u32 j = 0;
      for (; j + 128 <= size; j += 128) {
        for (u32 i = 0; i < 128; i++) classes[k][j + i] = simd_get(payload, ex, k, i);
        ex += 4 * k;
      }
      for (; j < size; j += 32) {
        u32 tmp[32];
        const u32 used = std::min<u32>(32, size - j);
        // the last pack is cut to the words its used values touch
        const size_t need = ((size_t)used * k + 31) / 32;
        payload.at(ex * 4, need * 4);
        std::vector<u32> padded(k, 0);
        memcpy(padded.data(), payload.at(ex * 4, need * 4), need * 4);
        unpack32(Span{(const u8*)padded.data(), (size_t)k * 4, nullptr}, 0, k, tmp);
        for (u32 i = 0; i < used; i++) classes[k][j + i] = tmp[i];
        ex += k;
      }
      ex -= (size_t)(j - size) * k / 32;
    }
    size_t data = header + 1;
    while ((misalign + data) % 4) data++;
    size_t bp = 0;
    size_t next[33] = {0};
    for (u32 blk = 0; blk < len; blk += 256) {
      if (bp + 2 > bytesize) throw DecodeError("PFOR byte container too short");
      const u32 b = bytes[bp++], cexcept = bytes[bp++];
      if (b > 32) throw DecodeError("PFOR bit width above 32");
      u32* o = out + done + blk;
      for (u32 i = 0; i < 256; i++) o[i] = simd_get(payload, data + (i >> 7) * 4 * (size_t)b, b, i & 127);
      data += 8 * (size_t)b;
      if (cexcept > 0) {
        if (bp + 1 + cexcept > bytesize) throw DecodeError("PFOR byte container too short");
        const u32 maxbits = bytes[bp++];
        if (maxbits <= b || maxbits > 32) throw DecodeError("bad PFOR exception width");
        const u32 k = maxbits - b;
        for (u32 e = 0; e < cexcept; e++) {
          const u32 pos = bytes[bp++];
          if (k == 1) {
            o[pos] |= 1u << b;
          } else {
            if (next[k] >= classes[k].size()) throw DecodeError("PFOR exception stream too short");
            o[pos] |= classes[k][next[k]++] << b;
          }
        }
      }
    }
    if (data > meta) throw DecodeError("PFOR packed data overruns the metadata");
    in = ex;
    done += len;
  }
  if (in > words) throw DecodeError("PFOR payload overruns its word count");
  const size_t tail = vbyte_decode(payload, in, words, out + rounded, n - rounded);
  if (rounded + tail != n) throw DecodeError("FastPFor stream holds the wrong number of values");
}

// ------------------------------------------------------------------------------------------
// CRoaring native serialization (third-party; see the header comment)
// ------------------------------------------------------------------------------------------
// Calls fn(row) for every member in ascending order, like Roaring::iterate. Returns cardinality.
u32 roaring_iterate(const Span& s, const std::function<void(u32)>& fn) {
  const u8 tag = s.b(0);
  if (tag == 1) {  // SERIALIZATION_ARRAY_UINT32: u32 cardinality, then that many u32 values
    const u32 card = s.w(1);
    for (u32 i = 0; i < card; i++) fn(s.w(5 + 4ull * i));
    return card;
  }
  if (tag != 2) throw DecodeError("unknown Roaring serialization tag");
  const Span r = s.from(1);  // portable RoaringFormatSpec stream
  const u32 cookie = r.w(0);
  size_t pos;
  u32 size;
  const u8* run_flags = nullptr;
  bool has_run = false;
  if ((cookie & 0xFFFF) == 12347) {
    size = (cookie >> 16) + 1;
    has_run = true;
    run_flags = r.at(4, (size + 7) / 8);
    pos = 4 + (size + 7) / 8;
  } else if (cookie == 12346) {
    size = r.w(4);
    pos = 8;
  } else {
    throw DecodeError("bad Roaring cookie");
  }
  if (size > 65536) throw DecodeError("bad Roaring container count");
  const size_t desc = pos;
  pos += 4ull * size;
  if (!has_run || size >= 4) pos += 4ull * size;  // offset header, unused when reading in order
  u32 total = 0;
  for (u32 c = 0; c < size; c++) {
    const u32 key = r.h(desc + 4ull * c);
    const u32 card = (u32)r.h(desc + 4ull * c + 2) + 1;
    const u32 hi = key << 16;
    const bool is_run = has_run && (run_flags[c / 8] >> (c % 8) & 1);
    if (is_run) {
      const u32 n_runs = r.h(pos);
      pos += 2;
      for (u32 k = 0; k < n_runs; k++, pos += 4) {
        const u32 start = r.h(pos), len = r.h(pos + 2);
        for (u32 v = start; v <= start + len; v++) fn(hi | v);
        total += len + 1;
      }
    } else if (card <= 4096) {
      for (u32 k = 0; k < card; k++, pos += 2) fn(hi | r.h(pos));
      total += card;
    } else {
      for (u32 wd = 0; wd < 1024; wd++, pos += 8) {
        u64 bits = r.q(pos);
        while (bits) {
          fn(hi | (wd * 64 + (u32)__builtin_ctzll(bits)));
          bits &= bits - 1;
          total++;
        }
      }
    }
  }
  return total;
}

// ------------------------------------------------------------------------------------------
// schemes
// ------------------------------------------------------------------------------------------
struct NullmapView {  // what FOR sees of the chunk's nullmap (integer/FOR.cpp:56)
  bool present;
  u8 type;
};

void decode_int(s32* dest, const NullmapView& nm, const Span& src, u32 n, u8 scheme, int depth);
void decode_dbl(double* dest, const Span& src, u32 n, u8 scheme, int depth);

constexpr int kMaxDepth = 16;

// XPBPStructure{u32 u32_count; u8 padding; data@5} -> FastPFor payload at data+padding
// (integer/PBP.hpp:7-12, PBP.cpp:145-157)
void decode_bp(u32* dest, const Span& src, size_t n) {
  const u32 words = src.w(0);
  const u32 padding = src.b(4);
  fbp_decode(src.from(5 + padding), words, dest, n);
}

// templated/RLE.hpp:133-193 (int) and :196-250 (double): decode values and counts, then expand.
template <typename T>
void decode_rle(T* dest, const Span& src, u32 n, int depth) {
  const u32 runs = src.w(0);
  const u32 counts_off = src.w(4);
  const u8 values_scheme = src.b(8), counts_scheme = src.b(9);
  if (runs > n) throw DecodeError("RLE run count above tuple count");
  std::vector<T> values(runs);
  std::vector<s32> counts(runs);
  const Span data = src.from(10);
  if constexpr (sizeof(T) == 4)
    decode_int((s32*)values.data(), NullmapView{false, 0}, data, runs, values_scheme, depth + 1);
  else
    decode_dbl((double*)values.data(), data, runs, values_scheme, depth + 1);
  decode_int(counts.data(), NullmapView{false, 0}, data.from(counts_off), runs, counts_scheme,
             depth + 1);
  u64 w = 0;
  for (u32 r = 0; r < runs; r++) {
    if (counts[r] <= 0 || w + (u64)counts[r] > n) throw DecodeError("RLE run lengths do not add up");
    for (s32 k = 0; k < counts[r]; k++) dest[w++] = values[r];
  }
  if (w != n) throw DecodeError("RLE run lengths do not add up");
}

// templated/DynamicDictionary.hpp:108-159 (int), :162-216 (double): dest[i] = dict[codes[i]];
// packed header {u8 codes_scheme; u32 codes_offset}, dictionary at +5.
template <typename T>
void decode_dict(T* dest, const Span& src, u32 n, int depth) {
  const u8 codes_scheme = src.b(0);
  const u32 codes_off = src.w(1);
  const Span data = src.from(5);
  std::vector<s32> codes(n);
  decode_int(codes.data(), NullmapView{false, 0}, data.from(codes_off), n, codes_scheme, depth + 1);
  const u32 dict_count = codes_off / sizeof(T);
  const u8* dict = data.at(0, (size_t)dict_count * sizeof(T));
  for (u32 i = 0; i < n; i++) {
    if ((u32)codes[i] >= dict_count) throw DecodeError("dictionary code out of range");
    memcpy(&dest[i], dict + (size_t)(u32)codes[i] * sizeof(T), sizeof(T));
  }
}

// templated/Frequency.hpp:99-136: fill with top_value, then the k-th bitmap member (ascending)
// receives exceptions[k]. Header {T top_value; u32 exceptions_offset; u8 next_scheme}, data at
// +9 (int) / +13 (double).
template <typename T>
void decode_frequency(T* dest, const Span& src, u32 n, int depth) {
  constexpr size_t kExcOff = sizeof(T), kNext = sizeof(T) + 4, kData = sizeof(T) + 5;
  T top;
  memcpy(&top, src.at(0, sizeof(T)), sizeof(T));
  const u32 exc_off = src.w(kExcOff);
  const u8 next = src.b(kNext);
  const Span data = src.from(kData);
  std::vector<u32> rows;
  roaring_iterate(data, [&](u32 r) { rows.push_back(r); });
  std::vector<T> exceptions(rows.size());
  if (!rows.empty()) {
    if constexpr (sizeof(T) == 4)
      decode_int((s32*)exceptions.data(), NullmapView{false, 0}, data.from(exc_off),
                 (u32)rows.size(), next, depth + 1);
    else
      decode_dbl((double*)exceptions.data(), data.from(exc_off), (u32)rows.size(), next, depth + 1);
  }
  for (u32 i = 0; i < n; i++) dest[i] = top;
  for (size_t k = 0; k < rows.size(); k++) {
    if (rows[k] >= n) throw DecodeError("Frequency exception row out of range");
    dest[rows[k]] = exceptions[k];
  }
}

// templated/FixedDictionary.hpp:56-71: {u32 codes_offset; T dict[]}, raw u8/u16 codes at
// src+codes_offset.
template <typename T, typename Code>
void decode_fixed_dict(T* dest, const Span& src, u32 n) {
  const u32 codes_off = src.w(0);
  const size_t slots = sizeof(T) == 4 ? 4 : 8;
  if (codes_off < slots) throw DecodeError("bad fixed dictionary header");
  const u32 dict_count = (u32)((codes_off - slots) / sizeof(T));
  const u8* dict = src.at(slots, (size_t)dict_count * sizeof(T));
  const u8* codes = src.at(codes_off, (size_t)n * sizeof(Code));
  for (u32 i = 0; i < n; i++) {
    Code c;
    memcpy(&c, codes + (size_t)i * sizeof(Code), sizeof(Code));
    if (c >= dict_count) throw DecodeError("dictionary code out of range");
    memcpy(&dest[i], dict + (size_t)c * sizeof(T), sizeof(T));
  }
}

void decode_int(s32* dest, const NullmapView& nm, const Span& src, u32 n, u8 scheme, int depth) {
  if (depth > kMaxDepth) throw DecodeError("cascade too deep");
  switch (scheme) {
    case 0:  // UNCOMPRESSED, integer/Uncompressed.cpp:26-33
      if (n) memcpy(dest, src.at(0, (size_t)n * 4), (size_t)n * 4);
      return;
    case 1: {  // ONE_VALUE, integer/OneValue.cpp:30-39
      const s32 v = (s32)src.w(0);
      for (u32 i = 0; i < n; i++) dest[i] = v;
      return;
    }
    case 2: decode_dict(dest, src, n, depth); return;
    case 3: decode_rle(dest, src, n, depth); return;
    case 5: decode_bp((u32*)dest, src, n); return;
    case 25: decode_frequency(dest, src, n, depth); return;
    case 26: {  // FOR, integer/FOR.cpp:46-65: child decode (nullmap handed down), then add bias
      const s32 bias = (s32)src.w(0);
      const u8 next = src.b(4);
      decode_int(dest, nm, src.from(5), n, next, depth + 1);
      if (nm.present && nm.type == ALLZEROS) return;
      for (u32 i = 0; i < n; i++) dest[i] = (s32)((u32)dest[i] + (u32)bias);
      return;
    }
    case 30: decode_fixed_dict<s32, u8>(dest, src, n); return;
    case 31: decode_fixed_dict<s32, u16>(dest, src, n); return;
    case 4:     // PFOR, integer/PBP.cpp:37-49
    case 27: {  // PFOR_DELTA, integer/PBP.cpp:92-109: PFOR, then the prefix sum of revertDelta
      const u32 words = src.w(0);
      const u32 padding = src.b(4);
      const Span payload = src.from(5 + padding);
      if (!payload.origin) throw DecodeError("PFOR needs the chunk origin");
      const size_t off = (size_t)(payload.p - payload.origin);
      if (off % 4) throw DecodeError("PFOR payload is not word aligned");
      pfor_decode(payload, words, (u32*)dest, n, (u32)((off / 4) % 4));
      if (scheme == 27)
        for (u32 i = 1; i < n; i++) dest[i] = (s32)((u32)dest[i] + (u32)dest[i - 1]);
      return;
    }
    default:
      // Truncation (28/29) is UNREACHABLE in the reference too (integer/Truncation.hpp:88-97).
      throw DecodeError("integer scheme cannot be decoded");
  }
}

// double/Pseudodecimal.cpp:19-43
const double kFractions[32] = {
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

// double/Pseudodecimal.cpp:348-452. The Roaring map of 4-row groups and the variant bits only
// steer the reference's AVX strategy; the result equals the scalar loop (:424-436), which is
// what is restated here. An exponent index above 22 reads past the reference's 23-entry table
// (undefined there); it is rejected here.
void decode_pseudodecimal(double* dest, const Span& src, u32 n, int depth) {
  const u32 converted = src.w(0);
  const u8 numbers_scheme = src.b(4), exponents_scheme = src.b(5), patches_scheme = src.b(6);
  const u32 exponents_off = src.w(8), patches_off = src.w(12);
  if (converted > n) throw DecodeError("Pseudodecimal converted_count above tuple count");
  const Span data = src.from(20);
  std::vector<s32> numbers(converted), exponents(n);
  std::vector<double> patches(n - converted);
  if (converted > 0)
    decode_int(numbers.data(), NullmapView{false, 0}, data, converted, numbers_scheme, depth + 1);
  decode_int(exponents.data(), NullmapView{false, 0}, data.from(exponents_off), n, exponents_scheme,
             depth + 1);
  decode_dbl(patches.data(), data.from(patches_off), n - converted, patches_scheme, depth + 1);
  size_t ni = 0, pi = 0;
  for (u32 i = 0; i < n; i++) {
    const s32 e = exponents[i];
    if (e == 23) {
      if (pi == patches.size()) throw DecodeError("Pseudodecimal has more exceptions than patches");
      dest[i] = patches[pi++];
    } else {
      if (ni == numbers.size()) throw DecodeError("Pseudodecimal has more digits rows than digits");
      const u32 idx = (u32)e & 0x1F;
      if (idx > 22) throw DecodeError("Pseudodecimal exponent out of range");
      dest[i] = (double)numbers[ni++] * kFractions[idx];
    }
  }
}

void decode_dbl(double* dest, const Span& src, u32 n, u8 scheme, int depth) {
  if (depth > kMaxDepth) throw DecodeError("cascade too deep");
  switch (scheme) {
    case 0:  // UNCOMPRESSED, double/Uncompressed.cpp:25-31
      if (n) memcpy(dest, src.at(0, (size_t)n * 8), (size_t)n * 8);
      return;
    case 1: {  // ONE_VALUE, double/OneValue.cpp:36-41
      double v;
      memcpy(&v, src.at(0, 8), 8);
      for (u32 i = 0; i < n; i++) dest[i] = v;
      return;
    }
    case 2: decode_dict(dest, src, n, depth); return;
    case 3: decode_rle(dest, src, n, depth); return;
    case 4: decode_frequency(dest, src, n, depth); return;
    case 5: decode_pseudodecimal(dest, src, n, depth); return;
    case 28:  // DOUBLE_BP, double/DoubleBP.cpp:41-57: FastBinaryPacking over 2n words
      decode_bp((u32*)dest, src, (size_t)n * 2);
      return;
    case 29: decode_fixed_dict<double, u8>(dest, src, n); return;
    case 31: decode_fixed_dict<double, u16>(dest, src, n); return;
    default: throw DecodeError("double scheme cannot be decoded");
  }
}

// extern/RoaringBitmap.cpp:32-114: ALLONES / ALLZEROS / REGULAR (members are valid rows) /
// FLIPPED (members are null rows) -> one byte per row.
void decode_nullmap(u8* dest, const Span& src, u8 type, u32 n) {
  switch (type) {
    case ALLONES: memset(dest, 1, n); return;
    case ALLZEROS: memset(dest, 0, n); return;
    case REGULAR:
    case FLIPPED: {
      const u8 member = type == REGULAR ? 1 : 0;
      memset(dest, 1 - member, n);
      roaring_iterate(src, [&](u32 r) {
        if (r >= n) throw DecodeError("nullmap row out of range");
        dest[r] = member;
      });
      return;
    }
    default: throw DecodeError("unknown nullmap type");
  }
}

// compression/BtrReader.cpp:31-74 (readColumn) over one ColumnChunkMeta chunk
// (compression/Datablock.hpp:12-21).
void decode_chunk(const u8* chunk, size_t bytes, void* dst, size_t dst_bytes, u8* nullmap_dst) {
  const Span s{chunk, bytes, chunk};
  const u8 scheme = s.b(0), nm_type = s.b(1), col = s.b(2);
  const u32 nm_off = s.w(4), n = s.w(8);
  const Span data = s.from(12);
  if (col == 0) {
    if (dst_bytes < (size_t)n * 4) throw DecodeError("destination too small");
    decode_int((s32*)dst, NullmapView{true, nm_type}, data, n, scheme, 0);
  } else if (col == 1) {
    if (dst_bytes < (size_t)n * 8) throw DecodeError("destination too small");
    decode_dbl((double*)dst, data, n, scheme, 0);
  } else {
    throw DecodeError("only integer and double columns are supported");
  }
  if (nullmap_dst) decode_nullmap(nullmap_dst, nm_type >= REGULAR ? data.from(nm_off) : data, nm_type, n);
}

// Integer sub-streams of a string column chunk in the DICT scheme. Header:
// scheme/string/DynamicDictionary.hpp:7-18 (not packed: use_fsst is followed by 3 padding bytes,
// data starts at byte 27; every *_offset counts from data).
//   which 0  codes of all rows:        codes_scheme.decompress, DynamicDictionary.cpp:381-387
//   which 1  lengths of all entries:   lengths_scheme.decompress, :444-451 (FSST chunks only)
//   which 2  run values, which 3  run lengths of the RLE-optimised path:
//            rle.decompressRuns, :239-251 / :484-496 and templated/RLE.hpp:111-128
struct StringDict {
  u32 n, total, num_codes;
  bool use_fsst, use_rle;
  u8 codes_scheme, lengths_scheme;
  Span data;
  Span codes, lengths;
};
StringDict open_string_dict(const u8* chunk, size_t bytes) {
  const Span s{chunk, bytes, chunk};
  if (s.b(2) != 2 || s.b(0) != 2) throw DecodeError("not a string column chunk in the DICT scheme");
  StringDict d;
  d.n = s.w(8);
  const Span h = s.from(12);
  d.total = h.w(0);
  d.use_fsst = h.b(4) != 0;
  d.num_codes = h.w(16);
  d.use_rle = h.b(24) != 0;
  d.codes_scheme = h.b(25);
  d.lengths_scheme = h.b(26);
  d.data = h.from(27);
  d.codes = d.data.from(h.w(20));
  d.lengths = d.use_fsst ? d.data.from(h.w(12)) : d.data;
  return d;
}
void decode_string_dict_stream(const u8* chunk, size_t bytes, int which, std::vector<s32>& out) {
  const StringDict d = open_string_dict(chunk, bytes);
  const NullmapView none{false, 0};
  if (which == 0) {
    out.resize(d.n);
    decode_int(out.data(), none, d.codes, d.n, d.codes_scheme, 1);
  } else if (which == 1) {
    if (!d.use_fsst) throw DecodeError("a chunk without FSST has no lengths stream");
    if (d.num_codes > d.n) throw DecodeError("more dictionary entries than rows");
    out.resize(d.num_codes);
    decode_int(out.data(), none, d.lengths, d.num_codes, d.lengths_scheme, 1);
  } else if (which == 2 || which == 3) {
    if (!d.use_rle || d.codes_scheme != 3) throw DecodeError("the chunk does not use the RLE-optimised path");
    const u32 runs = d.codes.w(0), counts_off = d.codes.w(4);
    if (runs > d.n) throw DecodeError("RLE run count above tuple count");
    const Span rle_data = d.codes.from(10);
    out.resize(runs);
    if (which == 2)
      decode_int(out.data(), none, rle_data, runs, d.codes.b(8), 2);
    else
      decode_int(out.data(), none, rle_data.from(counts_off), runs, d.codes.b(9), 2);
  } else {
    throw DecodeError("unknown sub-stream");
  }
}

// Rows of a DICT chunk without FSST: the copy loop of DynamicDictionary.cpp:405-416 over the
// StringArrayViewer dictionary (slot i = offset of string i from the first slot).
void decode_string_dict_rows(const u8* chunk, size_t bytes, std::vector<u32>& offsets, std::vector<u8>& text) {
  const StringDict d = open_string_dict(chunk, bytes);
  if (d.use_fsst) throw DecodeError("FSST dictionaries are not decoded");
  std::vector<s32> codes(d.n);
  decode_int(codes.data(), NullmapView{false, 0}, d.codes, d.n, d.codes_scheme, 1);
  offsets.assign((size_t)d.n + 1, 0);
  text.clear();
  for (u32 i = 0; i < d.n; i++) {
    if (codes[i] < 0 || (u32)codes[i] >= d.num_codes) throw DecodeError("dictionary code out of range");
    const u32 a = d.data.w(4ull * codes[i]), b = d.data.w(4ull * codes[i] + 4);
    if (b < a) throw DecodeError("dictionary slots decrease");
    const u8* p = d.data.at(a, b - a);
    offsets[i] = (u32)text.size();
    text.insert(text.end(), p, p + (b - a));
  }
  offsets[d.n] = (u32)text.size();
}

}  // namespace

extern "C" {

static thread_local std::string g_err;

const char* btr_oracle_error() { return g_err.c_str(); }

// Decodes one column chunk. Returns 0, or -1 with btr_oracle_error() set.
int btr_oracle_decode_chunk(const uint8_t* chunk, size_t bytes, void* dst, size_t dst_bytes,
                            uint8_t* nullmap_dst) {
  try {
    decode_chunk(chunk, bytes, dst, dst_bytes, nullmap_dst);
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

int btr_oracle_chunk_info(const uint8_t* chunk, size_t bytes, uint32_t* tuple_count,
                          uint8_t* col_type, uint8_t* scheme, uint8_t* nullmap_type) {
  if (bytes < 12) return -1;
  *scheme = chunk[0];
  *nullmap_type = chunk[1];
  *col_type = chunk[2];
  memcpy(tuple_count, chunk + 8, 4);
  return 0;
}

// Datablock::decompress (compression/Datablock.cpp:340-417) for one column of a relation block.
int btr_oracle_decode_datablock_column(const uint8_t* block, size_t bytes, uint32_t column,
                                       void* dst, size_t dst_bytes, uint8_t* nullmap_dst) {
  try {
    const Span s{block, bytes, block};
    const u32 n = s.w(0), cols = s.w(8);
    if (column >= cols) throw DecodeError("column index out of range");
    const size_t m = 16 + 16ull * column;
    const u8 scheme = s.b(m), col = s.b(m + 1), nm_type = s.b(m + 2);
    const u32 off = s.w(m + 4), nm_off = s.w(m + 12);
    if (col == 0) {
      if (dst_bytes < (size_t)n * 4) throw DecodeError("destination too small");
      decode_int((s32*)dst, NullmapView{true, nm_type}, s.from(off), n, scheme, 0);
    } else if (col == 1) {
      if (dst_bytes < (size_t)n * 8) throw DecodeError("destination too small");
      decode_dbl((double*)dst, s.from(off), n, scheme, 0);
    } else {
      throw DecodeError("only integer and double columns are supported");
    }
    if (nullmap_dst) decode_nullmap(nullmap_dst, s.from(nm_type >= REGULAR ? nm_off : 0), nm_type, n);
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// One integer sub-stream of a string DICT chunk (see decode_string_dict_stream). Writes at most
// cap values; *count receives the length of the stream.
int btr_oracle_string_dict_stream(const uint8_t* chunk, size_t bytes, int which, int32_t* out, size_t cap,
                                  uint32_t* count) {
  try {
    std::vector<s32> v;
    decode_string_dict_stream(chunk, bytes, which, v);
    if (count) *count = (u32)v.size();
    if (v.size() > cap) throw DecodeError("destination too small");
    if (!v.empty()) memcpy(out, v.data(), v.size() * 4);
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// Rows of a string DICT chunk without FSST: offsets[n + 1] into text. *text_bytes receives the
// total length; text may be NULL to query it.
int btr_oracle_string_dict_rows(const uint8_t* chunk, size_t bytes, uint32_t* offsets, size_t offsets_cap,
                                uint8_t* text, size_t text_cap, uint64_t* text_bytes) {
  try {
    std::vector<u32> off;
    std::vector<u8> txt;
    decode_string_dict_rows(chunk, bytes, off, txt);
    if (text_bytes) *text_bytes = txt.size();
    if (text) {
      if (off.size() > offsets_cap || txt.size() > text_cap) throw DecodeError("destination too small");
      memcpy(offsets, off.data(), off.size() * 4);
      if (!txt.empty()) memcpy(text, txt.data(), txt.size());
    }
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// Restatement of tools/conversion/decompression-speed.cpp:40-64,173-185,205-238: one task per
// chunk, a thread-local output buffer per worker that only grows, one untimed warm-up pass and
// `reps` timed passes; std::thread workers pulling chunk indices replace TBB. Nullmaps are
// re-expanded every pass (the reference resets its bitmap cache per pass, :40-43).
// Returns the mean seconds per pass (or a negative value on a decode error); *out_bytes receives
// the decompressed bytes of one pass.
double btr_oracle_decompression_speed(const uint8_t* const* chunks, const size_t* sizes, size_t n,
                                      int threads, int reps, uint64_t* out_bytes) {
  if (threads < 1) threads = (int)std::thread::hardware_concurrency();
  if (threads < 1) threads = 1;
  std::atomic<bool> failed{false};
  std::atomic<uint64_t> bytes_total{0};
  std::vector<std::vector<u8>> outs(threads);  // one growing output buffer per worker
  auto pass = [&](bool count) {
    std::atomic<size_t> next{0};
    auto worker = [&](int tid) {
      std::vector<u8>& out = outs[tid];
      uint64_t local = 0;
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= n) break;
        if (sizes[i] < 12) {
          failed = true;
          continue;
        }
        u32 tuples;
        memcpy(&tuples, chunks[i] + 8, 4);
        const size_t need = (size_t)tuples * (chunks[i][2] == 1 ? 8 : 4);
        if (out.size() < need) out.resize(need);
        try {
          decode_chunk(chunks[i], sizes[i], out.data(), out.size(), nullptr);
          if (chunks[i][1] >= REGULAR) {  // BtrReader::getBitmap parses the Roaring nullmap
            u32 nm_off;
            memcpy(&nm_off, chunks[i] + 4, 4);
            roaring_iterate(Span{chunks[i], sizes[i]}.from(12).from(nm_off), [](u32) {});
          }
        } catch (const std::exception&) {
          failed = true;
        }
        local += need;
      }
      if (count) bytes_total += local;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
  };
  pass(false);
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) pass(r == 0);
  const auto t1 = std::chrono::steady_clock::now();
  if (out_bytes) *out_bytes = bytes_total.load();
  if (failed) return -1.0;
  return std::chrono::duration<double>(t1 - t0).count() / (reps > 0 ? reps : 1);
}

int btr_oracle_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
