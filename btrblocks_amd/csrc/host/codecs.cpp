// CPU encoders for the FastPFor binary-packing payload and CRoaring's native serialization.
// See codecs.h for provenance.
#include "codecs.h"

#include <algorithm>
#include <cstring>
#include <stdexcept>

#include "../btr_format.h"

namespace btrhost {
namespace {

inline uint32_t bits_needed(uint32_t v) { return v == 0 ? 0 : 32u - (uint32_t)__builtin_clz(v); }

// 32 values, `b` bits each, concatenated LSB-first into b consecutive words.
void pack_miniblock(const uint32_t* in, uint32_t b, std::vector<uint32_t>& out) {
  if (b == 0) return;
  if (b == 32) {
    out.insert(out.end(), in, in + 32);
    return;
  }
  uint64_t acc = 0;
  uint32_t filled = 0;
  for (uint32_t i = 0; i < 32; i++) {
    acc |= (uint64_t)in[i] << filled;
    filled += b;
    if (filled >= 32) {
      out.push_back((uint32_t)acc);
      acc >>= 32;
      filled -= 32;
    }
  }
}

// VariableByte: 7 payload bits per byte, low group first, the LAST byte of a value has bit 7
// set; the byte stream is zero-padded to a word boundary.
void vbyte_encode(const uint32_t* in, size_t n, std::vector<uint32_t>& out) {
  std::vector<uint8_t> bytes;
  for (size_t i = 0; i < n; i++) {
    uint32_t v = in[i];
    while (v >= 128) {
      bytes.push_back((uint8_t)(v & 127));
      v >>= 7;
    }
    bytes.push_back((uint8_t)(v | 128));
  }
  while (bytes.size() % 4) bytes.push_back(0);
  size_t w0 = out.size();
  out.resize(w0 + bytes.size() / 4);
  if (!bytes.empty()) memcpy(out.data() + w0, bytes.data(), bytes.size());
}

}  // namespace

size_t fbp_encode(const uint32_t* in, size_t n, std::vector<uint32_t>& out) {
  const size_t start = out.size();
  const size_t rounded = n / btrfmt::FBP_MINIBLOCK * btrfmt::FBP_MINIBLOCK;
  out.push_back((uint32_t)rounded);
  for (size_t g = 0; g < rounded; g += btrfmt::FBP_GROUP) {
    const size_t minis = std::min<size_t>(4, (rounded - g) / btrfmt::FBP_MINIBLOCK);
    uint32_t b[4] = {0, 0, 0, 0};
    for (size_t m = 0; m < minis; m++) {
      uint32_t acc = 0;
      for (size_t i = 0; i < 32; i++) acc |= in[g + m * 32 + i];
      b[m] = bits_needed(acc);
    }
    out.push_back((b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3]);
    for (size_t m = 0; m < minis; m++) pack_miniblock(in + g + m * 32, b[m], out);
  }
  if (rounded < n) vbyte_encode(in + rounded, n - rounded, out);
  return out.size() - start;
}

// ------------------------------------------------------------------------------------------
// SIMDFastPFor<8> (restated from FastPFor's simdfastpfor.h / simdbitpacking):
//   word 0            number of values coded by this codec (a multiple of 256)
//   per page of at most 65536 values:
//     header word     distance in words from the header to the metadata
//     zero words up to a 16-byte boundary of the absolute address
//     per 256-value block: two 128-value SIMD packs of b words x 4 lanes each
//     metadata        u32 byte count; the byte container {b, exception count, [max bits, positions...]}
//                     per block, zero-padded to a word; u32 bitmap of the exception widths in use
//                     (bit k-1 for width k = max bits - b >= 2); per width in ascending order: u32
//                     count, then the values (value >> b) packed at that width - SIMD packs of 128,
//                     then 32-value horizontal packs, the last one cut to whole used words
//   A width-1 exception is implied (bit b set), it stores nothing.
// SIMD pack of 128 values at width b: value i lives in lane i % 4 at bit (i / 4) * b of that
// lane's stream; word k of lane L is word 4k + L of the pack.
namespace {

void simd_pack128(const uint32_t* in, uint32_t b, std::vector<uint32_t>& out) {
  const size_t base = out.size();
  out.resize(base + 4 * (size_t)b, 0);
  if (b == 0) return;
  for (uint32_t i = 0; i < 128; i++) {
    const uint32_t lane = i & 3, bit = (i >> 2) * b, k = bit >> 5, sh = bit & 31;
    const uint64_t v = (uint64_t)(b == 32 ? in[i] : (in[i] & ((1u << b) - 1))) << sh;
    out[base + 4 * k + lane] |= (uint32_t)v;
    if (sh + b > 32) out[base + 4 * (k + 1) + lane] |= (uint32_t)(v >> 32);
  }
}

void pack_exceptions(const std::vector<uint32_t>& src, uint32_t bit, std::vector<uint32_t>& out) {
  const size_t size = src.size();
  out.push_back((uint32_t)size);
  if (size == 0) return;
  std::vector<uint32_t> padded(src);
  padded.resize((size + 31) / 32 * 32, 0);
  size_t j = 0;
  for (; j + 128 <= size; j += 128) simd_pack128(padded.data() + j, bit, out);
  for (; j < size; j += 32) pack_miniblock(padded.data() + j, bit, out);
  out.resize(out.size() - (j - size) * bit / 32);  // whole words the padding alone would occupy
}

// Width b, exception count and largest width of one block: FastPFor's cost model (8 bits per
// exception position, the exception's extra bits, 8 bits for the max-bits byte).
void best_b(const uint32_t* in, uint32_t& bestb, uint32_t& bestc, uint32_t& maxb) {
  uint32_t freqs[33] = {0};
  for (uint32_t k = 0; k < btrfmt::PFOR_BLOCK; k++) freqs[bits_needed(in[k])]++;
  bestb = 32;
  while (bestb > 0 && freqs[bestb] == 0) bestb--;
  maxb = bestb;
  uint32_t bestcost = bestb * btrfmt::PFOR_BLOCK, cexcept = 0;
  bestc = 0;
  for (uint32_t b = bestb; b-- > 0;) {
    cexcept += freqs[b + 1];
    if (cexcept > 255) break;  // the count is stored in one byte
    uint32_t cost = cexcept * 8 + cexcept * (maxb - b) + b * btrfmt::PFOR_BLOCK + 8;
    if (maxb - b == 1) cost -= cexcept;
    if (cost < bestcost) {
      bestcost = cost;
      bestb = b;
      bestc = cexcept;
    }
  }
}

}  // namespace

size_t pfor_encode(const uint32_t* in, size_t n, std::vector<uint32_t>& out, size_t first_word_mod4) {
  const size_t start = out.size();
  const size_t rounded = n / btrfmt::PFOR_BLOCK * btrfmt::PFOR_BLOCK;
  out.push_back((uint32_t)rounded);
  for (size_t page = 0; page < rounded; page += btrfmt::PFOR_PAGE) {
    const size_t len = std::min<size_t>(btrfmt::PFOR_PAGE, rounded - page);
    const size_t header = out.size();
    out.push_back(0);
    while ((first_word_mod4 + (out.size() - start)) % 4) out.push_back(0);  // 16-byte boundary
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> classes[33];
    for (size_t blk = 0; blk < len; blk += btrfmt::PFOR_BLOCK) {
      const uint32_t* v = in + page + blk;
      uint32_t b, cexcept, maxb;
      best_b(v, b, cexcept, maxb);
      bytes.push_back((uint8_t)b);
      bytes.push_back((uint8_t)cexcept);
      if (cexcept > 0) {
        bytes.push_back((uint8_t)maxb);
        for (uint32_t k = 0; k < btrfmt::PFOR_BLOCK; k++)
          if (b < 32 && (v[k] >> b) != 0) {
            classes[maxb - b].push_back(v[k] >> b);
            bytes.push_back((uint8_t)k);
          }
      }
      simd_pack128(v, b, out);
      simd_pack128(v + 128, b, out);
    }
    out[header] = (uint32_t)(out.size() - header);
    out.push_back((uint32_t)bytes.size());
    const size_t bw = out.size();
    out.resize(bw + (bytes.size() + 3) / 4, 0);
    if (!bytes.empty()) memcpy(out.data() + bw, bytes.data(), bytes.size());
    uint32_t bitmap = 0;
    for (uint32_t k = 2; k <= 32; k++)
      if (!classes[k].empty()) bitmap |= 1u << (k - 1);
    out.push_back(bitmap);
    for (uint32_t k = 2; k <= 32; k++)
      if (!classes[k].empty()) pack_exceptions(classes[k], k, out);
  }
  if (rounded < n) vbyte_encode(in + rounded, n - rounded, out);
  return out.size() - start;
}

// ------------------------------------------------------------------------------------------
// CRoaring native serialization
// ------------------------------------------------------------------------------------------
namespace {
struct Container {
  uint16_t key;
  uint32_t card;
  uint32_t n_runs;
  const uint32_t* begin;  // ids of this container (full 32-bit values)
  enum Kind { ARRAY, BITSET, RUN } kind;
  size_t bytes() const {
    switch (kind) {
      case ARRAY: return 2u * card;
      case BITSET: return 8192;
      default: return 2u + 4u * n_runs;
    }
  }
};
}  // namespace

size_t roaring_serialize(const uint32_t* ids, size_t n, std::vector<uint8_t>& out) {
  const size_t start = out.size();
  std::vector<Container> cs;
  for (size_t i = 0; i < n;) {
    Container c;
    c.key = (uint16_t)(ids[i] >> 16);
    c.begin = ids + i;
    size_t j = i;
    uint32_t runs = 0;
    while (j < n && (ids[j] >> 16) == c.key) {
      if (j == i || ids[j] != ids[j - 1] + 1) runs++;
      j++;
    }
    c.card = (uint32_t)(j - i);
    c.n_runs = runs;
    // runOptimize(): a run container replaces the array/bitset one only when strictly smaller.
    const size_t run_bytes = 2u + 4u * runs;
    if (c.card <= btrfmt::ROAR_ARRAY_MAX_CARD)
      c.kind = run_bytes < 2u * c.card + 2u ? Container::RUN : Container::ARRAY;
    else
      c.kind = run_bytes < 8192u ? Container::RUN : Container::BITSET;
    cs.push_back(c);
    i = j;
  }
  const size_t size = cs.size();
  bool has_run = false;
  for (auto& c : cs) has_run |= c.kind == Container::RUN;
  size_t header;
  const bool with_offsets = !has_run || size >= btrfmt::ROAR_NO_OFFSET_THRESHOLD;
  if (has_run)
    header = 4 + (size + 7) / 8 + 4 * size + (with_offsets ? 4 * size : 0);
  else
    header = 8 + 8 * size;
  size_t portable = header;
  for (auto& c : cs) portable += c.bytes();
  const size_t as_array = 4 * n + 4;

  if (!(portable < as_array)) {
    out.resize(start + 1 + as_array);
    uint8_t* p = out.data() + start;
    p[0] = btrfmt::ROAR_TAG_ARRAY_U32;
    btrfmt::wr32(p + 1, (uint32_t)n);
    if (n) memcpy(p + 5, ids, 4 * n);
    return out.size() - start;
  }
  out.resize(start + 1 + portable);
  uint8_t* base = out.data() + start + 1;
  out[start] = btrfmt::ROAR_TAG_CONTAINER;
  uint8_t* p = base;
  if (has_run) {
    btrfmt::wr32(p, btrfmt::ROAR_COOKIE_RUNS | (uint32_t)((size - 1) << 16));
    p += 4;
    memset(p, 0, (size + 7) / 8);
    for (size_t i = 0; i < size; i++)
      if (cs[i].kind == Container::RUN) p[i / 8] |= (uint8_t)(1u << (i % 8));
    p += (size + 7) / 8;
  } else {
    btrfmt::wr32(p, btrfmt::ROAR_COOKIE_NORUNS);
    btrfmt::wr32(p + 4, (uint32_t)size);
    p += 8;
  }
  for (auto& c : cs) {
    btrfmt::wr16(p, c.key);
    btrfmt::wr16(p + 2, (uint16_t)(c.card - 1));
    p += 4;
  }
  if (with_offsets) {
    size_t off = header;
    for (auto& c : cs) {
      btrfmt::wr32(p, (uint32_t)off);
      p += 4;
      off += c.bytes();
    }
  }
  for (auto& c : cs) {
    switch (c.kind) {
      case Container::ARRAY:
        for (uint32_t i = 0; i < c.card; i++, p += 2) btrfmt::wr16(p, (uint16_t)c.begin[i]);
        break;
      case Container::BITSET:
        memset(p, 0, 8192);
        for (uint32_t i = 0; i < c.card; i++) {
          const uint32_t lo = c.begin[i] & 0xFFFF;
          p[lo >> 3] |= (uint8_t)(1u << (lo & 7));
        }
        p += 8192;
        break;
      case Container::RUN: {
        btrfmt::wr16(p, (uint16_t)c.n_runs);
        p += 2;
        uint32_t i = 0;
        while (i < c.card) {
          uint32_t j = i;
          while (j + 1 < c.card && c.begin[j + 1] == c.begin[j] + 1) j++;
          btrfmt::wr16(p, (uint16_t)c.begin[i]);
          btrfmt::wr16(p + 2, (uint16_t)(j - i));  // stored as length-1
          p += 4;
          i = j + 1;
        }
        break;
      }
    }
  }
  return out.size() - start;
}

}  // namespace btrhost
