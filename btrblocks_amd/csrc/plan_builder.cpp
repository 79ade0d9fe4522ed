// Host-side plan compiler (see plan_builder.h and plan.h).
//
// Header layouts followed, per scheme:
//   UNCOMPRESSED   btrblocks/scheme/integer/Uncompressed.cpp:26-33, double/Uncompressed.cpp:25-31
//   ONE_VALUE      btrblocks/scheme/integer/OneValue.cpp:30-39, double/OneValue.cpp:36-41
//   DICT           btrblocks/scheme/templated/DynamicDictionary.hpp:7-11, 108-216
//   RLE            btrblocks/scheme/templated/RLE.hpp:8-14, 133-250
//   BP             btrblocks/scheme/integer/PBP.hpp:7-12, PBP.cpp:145-157
//   FREQUENCY      btrblocks/scheme/templated/Frequency.hpp:12-18, 99-136
//   FOR            btrblocks/scheme/integer/FOR.hpp:8-12, FOR.cpp:46-65
//   PSEUDODECIMAL  btrblocks/scheme/double/Pseudodecimal.hpp:13-32, Pseudodecimal.cpp:348-452
//   DOUBLE_BP      btrblocks/scheme/double/DoubleBP.hpp:7-11, DoubleBP.cpp:41-57
//   DICTIONARY_8/16 btrblocks/scheme/templated/FixedDictionary.hpp:7-11, 56-71
//   nullmap        btrblocks/extern/RoaringBitmap.cpp:8-26, 32-114
#include "plan_builder.h"

#include <algorithm>
#include <functional>
#include <cstring>

#include "btr_format.h"

namespace btrplan {
using namespace btrfmt;

namespace {

constexpr int ST_CORRUPT = 2, ST_UNSUPPORTED = 3, ST_MISALIGNED = 6;
constexpr int kMaxDepth = 8;
// Optional tables (materialized children, dictionary copies) go to LDS while the workgroup's
// total stays under this; beyond it they live in the chunk's HBM scratch slice.
constexpr uint32_t kLdsSoftCapGeneric = 52 * 1024;  // three workgroups per CU
constexpr uint32_t kLdsSoftCapStream = 60 * 1024;   // two 512-thread workgroups per CU when a dictionary needs it
constexpr uint32_t kDictLdsMaxBytes = 32 * 1024;
constexpr uint32_t kMaxTupleCount = 1u << 24;

struct Builder {
  const ColumnSource& cs;
  std::vector<Step> bp_steps, steps;
  std::vector<Pipe> pipes;
  std::vector<Seq> seqs;
  std::vector<uint32_t> consts;        // constant words appended to the program
  std::vector<uint32_t> pfor_outputs;  // sequences filled by S_PFOR_DECODE steps
  std::vector<uint32_t> transient;     // tables consumed by the step right after the one that fills them
  uint64_t needed = 0;
  uint32_t max_depth = 0;

  [[noreturn]] void corrupt(const std::string& what) const { throw PlanError(ST_CORRUPT, what); }
  void need(size_t off, size_t len, const char* what) const {
    if (off > cs.bytes || len > cs.bytes - off)
      corrupt(std::string(what) + " runs past the end of the compressed data");
  }
  uint8_t b(size_t off) const {
    need(off, 1, "header");
    return cs.host[off];
  }
  uint32_t w(size_t off) const {
    need(off, 4, "header");
    return rd32(cs.host + off);
  }
  uint16_t h(size_t off) const {
    need(off, 2, "header");
    return rd16(cs.host + off);
  }

  uint32_t add_seq(uint8_t loc, uint32_t off, uint32_t count, uint8_t esize, bool must_lds = false) {
    Seq s{};
    s.off = off;
    s.count = count;
    s.loc = loc;
    s.esize = esize;
    s.must_lds = must_lds ? 1 : 0;
    seqs.push_back(s);
    return (uint32_t)seqs.size() - 1;
  }
  uint32_t add_pipe(uint8_t base, bool is_double, uint32_t count) {
    Pipe p{};
    p.base = base;
    p.is_double = is_double ? 1 : 0;
    p.count = count;
    for (auto& x : p.a) x = NONE;
    pipes.push_back(p);
    return (uint32_t)pipes.size() - 1;
  }
  void add_step(uint8_t kind, uint32_t a0 = NONE, uint32_t a1 = NONE, uint32_t a2 = NONE,
                uint32_t a3 = NONE, uint32_t a4 = NONE, uint32_t a5 = NONE) {
    Step s{};
    s.kind = kind;
    s.a[0] = a0, s.a[1] = a1, s.a[2] = a2, s.a[3] = a3, s.a[4] = a4, s.a[5] = a5;
    (kind == S_BP_INDEX ? bp_steps : steps).push_back(s);
  }

  // The sequence holding a pipe's values: the pipe's own source when it is a bare array,
  // otherwise a new table filled by a materialize step.
  uint32_t as_seq(uint32_t pipe, uint32_t rows_reading_it = 0) {
    const Pipe& p = pipes[pipe];
    if (p.base == B_SEQ && p.n_post == 0) {
      // a raw array inside the compressed bytes that every row gathers from (run values,
      // exception lists): worth a copy into LDS when it is small against the rows it serves
      const Seq& s = seqs[p.a[0]];
      if (s.loc != LOC_SRC || rows_reading_it < 2 * s.count || (uint64_t)s.count * s.esize > kDictLdsMaxBytes ||
          s.count == 0)
        return p.a[0];
      const uint32_t lds = add_seq(LOC_LDS, 0, s.count, s.esize);
      add_step(S_MATERIALIZE, pipe, lds);
      return lds;
    }
    const uint32_t s = add_seq(LOC_LDS, 0, p.count, p.is_double ? 8 : 4);
    add_step(S_MATERIALIZE, pipe, s);
    return s;
  }
  uint32_t append_post(uint32_t pipe, uint8_t type, uint32_t arg) {
    if (pipes[pipe].n_post == MAX_POST) {
      const uint32_t s = as_seq(pipe);
      pipe = add_pipe(B_SEQ, false, seqs[s].count);
      pipes[pipe].a[0] = s;
    }
    Pipe& p = pipes[pipe];
    p.post[p.n_post].type = type;
    p.post[p.n_post].arg = arg;
    p.n_post++;
    return pipe;
  }

  // Whether the stream kernel can read this exponent pipe: bare, or behind one dictionary of at
  // most 32 entries that are all exponents the decoder accepts (0..22 and the exception code 23).
  // The kernel reads such a dictionary from the compressed bytes itself: its LDS copy and the
  // step that fills it are dropped.
  bool stream_exponents(uint32_t pipe, const uint8_t* host) {
    Pipe& p = pipes[pipe];
    if (p.n_post == 0) return true;
    if (p.n_post != 1 || p.post[0].type != POST_DICT) return false;
    uint32_t dict = p.post[0].arg;
    size_t copy_step = steps.size();
    if (seqs[dict].loc != LOC_SRC) {
      for (size_t k = 0; k < steps.size(); k++)
        if (steps[k].kind == S_MATERIALIZE && steps[k].a[1] == dict) copy_step = k;
      if (copy_step == steps.size()) return false;
      const Pipe& from = pipes[steps[copy_step].a[0]];
      if (from.base != B_SEQ || from.n_post != 0 || seqs[from.a[0]].loc != LOC_SRC) return false;
      dict = from.a[0];
    }
    const Seq& d = seqs[dict];
    if (d.esize != 4 || d.count == 0 || d.count > DEC_INDEX_MASK + 1) return false;
    for (uint32_t k = 0; k < d.count; k++) {
      uint32_t e;
      memcpy(&e, host + d.off + 4 * k, 4);
      if (e > DEC_EXCEPTION_CODE) return false;
    }
    if (copy_step != steps.size()) {
      seqs[p.post[0].arg].count = 0;
      steps.erase(steps.begin() + copy_step);
      p.post[0].arg = dict;
    }
    return true;
  }

  struct Bitmap {
    uint32_t bits, rank;
  };
  Bitmap add_bitmap(uint32_t nbits) {
    const uint32_t words = nbits / 64 + 1;  // one spare word keeps rank(nbits) addressable
    Bitmap bm;
    bm.bits = add_seq(LOC_LDS, 0, words, 8, true);
    bm.rank = add_seq(LOC_LDS, 0, words, 4, true);
    return bm;
  }

  // Parses CRoaring's native serialization at `off`, emits the steps that expand it into `bm`
  // and returns the cardinality. `count_bytes`: whether the decoder's reads count as needed.
  uint32_t roaring_steps(size_t off, const Bitmap& bm, uint32_t nbits, bool count_bytes) {
    add_step(S_BITMAP_CLEAR, NONE, bm.bits);
    const uint8_t tag = b(off);
    uint64_t used = 1;
    uint32_t card_total = 0;
    if (tag == ROAR_TAG_ARRAY_U32) {
      const uint32_t card = w(off + 1);
      if (card > nbits) corrupt("Roaring array holds more members than rows");
      need(off + 5, (size_t)card * 4, "Roaring array");
      add_step(S_BITMAP_SET, (uint32_t)(off + 5), bm.bits, card, 0, nbits, SET_ARRAY32);
      used += 4 + (uint64_t)card * 4;
      card_total = card;
    } else if (tag == ROAR_TAG_CONTAINER) {
      const size_t r = off + 1;
      const uint32_t cookie = w(r);
      uint32_t size;
      size_t pos;
      bool has_run = false;
      size_t flags = 0;
      if ((cookie & 0xFFFF) == ROAR_COOKIE_RUNS) {
        size = (cookie >> 16) + 1;
        has_run = true;
        flags = r + 4;
        need(flags, (size + 7) / 8, "Roaring run flags");
        pos = r + 4 + (size + 7) / 8;
      } else if (cookie == ROAR_COOKIE_NORUNS) {
        size = w(r + 4);
        pos = r + 8;
      } else {
        corrupt("bad Roaring cookie");
      }
      if (size > 65536) corrupt("bad Roaring container count");
      const size_t desc = pos;
      need(desc, (size_t)size * 4, "Roaring descriptive header");
      pos += (size_t)size * 4;
      if (!has_run || size >= ROAR_NO_OFFSET_THRESHOLD) pos += (size_t)size * 4;
      for (uint32_t c = 0; c < size; c++) {
        const uint32_t key = h(desc + 4 * (size_t)c);
        const uint32_t card = (uint32_t)h(desc + 4 * (size_t)c + 2) + 1;
        const uint32_t first_row = key << 16;
        if (first_row >= nbits) corrupt("Roaring container beyond the row range");
        const bool is_run = has_run && ((cs.host[flags + c / 8] >> (c % 8)) & 1);
        if (is_run) {
          const uint32_t n_runs = h(pos);
          need(pos + 2, (size_t)n_runs * 4, "Roaring run container");
          add_step(S_BITMAP_SET, (uint32_t)(pos + 2), bm.bits, n_runs, first_row, nbits, SET_RUNS);
          pos += 2 + (size_t)n_runs * 4;
        } else if (card <= ROAR_ARRAY_MAX_CARD) {
          need(pos, (size_t)card * 2, "Roaring array container");
          add_step(S_BITMAP_SET, (uint32_t)pos, bm.bits, card, first_row, nbits, SET_ARRAY16);
          pos += (size_t)card * 2;
        } else {
          need(pos, 8192, "Roaring bitset container");
          add_step(S_BITMAP_SET, (uint32_t)pos, bm.bits, 1024, first_row, nbits, SET_BITSET);
          pos += 8192;
        }
        card_total += card;
      }
      used = pos - off;
    } else {
      corrupt("unknown Roaring serialization tag");
    }
    if (card_total > nbits) corrupt("Roaring bitmap holds more members than rows");
    add_step(S_BITMAP_RANK, NONE, bm.bits, bm.rank, nbits, card_total, ST_ROARING);
    if (count_bytes) needed += used;
    return card_total;
  }

  // FastBinaryPacking<32> + VariableByte stream of `values` 32-bit values.
  uint32_t bp_node(size_t off, uint32_t rows, uint32_t values, bool is_double, std::string& desc) {
    const uint32_t words = w(off + BP_WORDS);
    const uint32_t padding = b(off + BP_PADDING);
    const size_t payload = off + BP_DATA + padding;
    need(payload, (size_t)words * 4, "bit-packed payload");
    if ((cs.dev_addr + payload) % 4 != 0)
      throw PlanError(ST_MISALIGNED,
                      "bit-packed payload is not 4-byte aligned at its device address; place the "
                      "compressed bytes at the alignment they were compressed at (16 bytes)");
    const uint32_t p = add_pipe(is_double ? B_DBP : B_BP, is_double, rows);
    uint32_t rounded = 0;
    if (values > 0) {
      if (words == 0) corrupt("empty bit-packed payload");
      rounded = w(payload);
      if (rounded != values / FBP_MINIBLOCK * FBP_MINIBLOCK)
        corrupt("bit-packed length word does not match the value count");
    }
    const uint32_t groups = (rounded + FBP_GROUP - 1) / FBP_GROUP;
    Pipe& pp = pipes[p];
    pp.a[0] = (uint32_t)payload;
    pp.a[1] = words;
    pp.a[3] = rounded;
    pp.a[5] = groups;
    const uint32_t index = add_seq(LOC_LDS, 0, groups + 1, 8, true);
    const uint32_t tail = values > rounded ? add_seq(LOC_LDS, 0, FBP_MINIBLOCK, 4, true) : NONE;
    pipes[p].a[2] = index;
    pipes[p].a[4] = tail;
    add_step(S_BP_INDEX, p);
    needed += BP_DATA + (uint64_t)words * 4;
    desc = is_double ? "DOUBLE_BP" : "BP";
    return p;
  }

  // SIMDFastPFor<8> + VariableByte stream (integer/PBP.cpp:37-49, 92-109). The page structure
  // (a handful of words per 65536 values) is parsed here; the per-block byte container and the
  // packed data are read on the device. The stream is always decoded into a table.
  uint32_t pfor_node(size_t off, uint32_t n, bool delta, std::string& desc) {
    const uint32_t words = w(off + BP_WORDS);
    const uint32_t padding = b(off + BP_PADDING);
    const size_t payload = off + BP_DATA + padding;
    need(payload, (size_t)words * 4, "PFOR payload");
    if ((cs.dev_addr + payload) % 4 != 0)
      throw PlanError(ST_MISALIGNED,
                      "PFOR payload is not 4-byte aligned at its device address; place the compressed "
                      "bytes at the alignment they were compressed at (16 bytes)");
    const uint32_t misalign = (uint32_t)(((cs.dev_addr + payload) / 4) % 4);
    auto pw = [&](size_t word) {
      if (word >= words) corrupt("PFOR metadata runs past the payload");
      return w(payload + word * 4);
    };
    const uint32_t consts_at = (uint32_t)consts.size();
    uint32_t pages = 0;
    size_t in = 0;
    if (n > 0) {
      if (words == 0) corrupt("empty PFOR payload");
      const uint32_t rounded = pw(0);
      if (rounded != n / PFOR_BLOCK * PFOR_BLOCK) corrupt("PFOR length word does not match the value count");
      in = 1;
      for (uint32_t done = 0; done < rounded;) {
        const uint32_t len = std::min<uint32_t>(PFOR_PAGE, rounded - done);
        const size_t header = in;
        const size_t meta = header + pw(header);
        const uint32_t bytesize = pw(meta);
        if (bytesize > (uint64_t)(words - meta) * 4) corrupt("PFOR byte container runs past the payload");
        size_t ex = meta + 1 + (bytesize + 3) / 4;
        const uint32_t bitmap = pw(ex);
        ex++;
        size_t data = header + 1;
        while ((misalign + data) % 4) data++;
        uint32_t cls_word[33] = {0}, cls_size[33] = {0};
        for (uint32_t k = 2; k <= 32; k++) {
          if (!((bitmap >> (k - 1)) & 1)) continue;
          const uint32_t size = pw(ex);
          ex++;
          cls_word[k] = (uint32_t)ex;
          cls_size[k] = size;
          const uint64_t full = size / 128, rest = (size % 128 + 31) / 32;
          uint64_t used = full * 4 * k + rest * k;
          used -= (uint64_t)((full * 128 + rest * 32) - size) * k / 32;
          if (ex + used > words) corrupt("PFOR exception values run past the payload");
          ex += used;
        }
        consts.push_back(done);
        consts.push_back(len);
        consts.push_back((uint32_t)data);
        consts.push_back((uint32_t)((meta + 1) * 4));
        consts.push_back(bytesize);
        for (uint32_t k = 2; k <= 32; k++) consts.push_back(cls_word[k]);
        for (uint32_t k = 2; k <= 32; k++) consts.push_back(cls_size[k]);
        if (data > meta) corrupt("PFOR packed blocks overlap the metadata");
        pages++;
        in = ex;
        done += len;
      }
      if (in > words) corrupt("PFOR pages run past the payload");
    }
    consts.push_back((uint32_t)in);  // first word of the VariableByte tail
    const uint32_t out = add_seq(LOC_LDS, 0, n, 4);
    const uint32_t work = add_seq(LOC_LDS, 0, kPforWorkBytes / 16, 16, true);
    add_step(S_PFOR_DECODE, (uint32_t)payload, words, consts_at, pages | (delta ? PFOR_DELTA_BIT : 0u), out, work);
    pfor_outputs.push_back(out);
    needed += BP_DATA + (uint64_t)words * 4;
    desc = delta ? "PFOR_DELTA" : "PFOR";
    const uint32_t p = add_pipe(B_SEQ, false, n);
    pipes[p].a[0] = out;
    return p;
  }

  template <typename CodeT>
  uint32_t fixed_dict_node(size_t off, uint32_t n, bool is_double, std::string& desc) {
    const uint32_t esz = is_double ? 8 : 4;
    const uint32_t slots = is_double ? FDICT_D_SLOTS : FDICT_I_SLOTS;
    const uint32_t codes_off = w(off + FDICT_CODES_OFFSET);
    if (codes_off < slots) corrupt("bad fixed dictionary header");
    const uint32_t dict_count = (codes_off - slots) / esz;
    need(off + slots, (size_t)dict_count * esz, "fixed dictionary");
    need(off + codes_off, (size_t)n * sizeof(CodeT), "fixed dictionary codes");
    const uint32_t codes = add_seq(LOC_SRC, (uint32_t)(off + codes_off), n, sizeof(CodeT));
    const uint32_t dict = add_seq(LOC_SRC, (uint32_t)(off + slots), dict_count, (uint8_t)esz);
    needed += slots + (uint64_t)dict_count * esz + (uint64_t)n * sizeof(CodeT);
    uint32_t cp = add_pipe(B_SEQ, false, n);
    pipes[cp].a[0] = codes;
    desc = sizeof(CodeT) == 1 ? "DICTIONARY_8" : "DICTIONARY_16";
    if (!is_double) return append_post(cp, POST_DICT, dict);
    const uint32_t p = add_pipe(B_DICTD, true, n);
    pipes[p].a[0] = cp;
    pipes[p].a[1] = dict;
    return p;
  }

  // A dictionary small enough is staged in LDS; the copy is an ordinary materialize step.
  uint32_t dict_seq(size_t off, uint32_t count, uint32_t esz, uint32_t rows) {
    const uint32_t src = add_seq(LOC_SRC, (uint32_t)off, count, (uint8_t)esz);
    if (count == 0 || (uint64_t)count * esz > kDictLdsMaxBytes || rows < 2 * count) return src;
    const uint32_t p = add_pipe(B_SEQ, esz == 8, count);
    pipes[p].a[0] = src;
    const uint32_t lds = add_seq(LOC_LDS, 0, count, (uint8_t)esz);
    add_step(S_MATERIALIZE, p, lds);
    return lds;
  }

  template <bool IS_DOUBLE>
  uint32_t rle_node(size_t off, uint32_t n, int depth, std::string& desc) {
    const uint32_t runs = w(off + RLE_RUNS);
    const uint32_t counts_off = w(off + RLE_COUNTS_OFFSET);
    const uint8_t vs = b(off + RLE_VALUES_SCHEME), cs_code = b(off + RLE_COUNTS_SCHEME);
    if (runs == 0 || runs > n) corrupt("RLE run count out of range");
    const size_t data = off + RLE_DATA;
    need(data, counts_off, "RLE values");
    needed += RLE_DATA;
    std::string dv, dc;
    const uint32_t vp = IS_DOUBLE ? dbl_node(data, vs, runs, depth + 1, dv)
                                  : int_node(data, vs, runs, false, depth + 1, dv);
    const uint32_t cp = int_node(data + counts_off, cs_code, runs, false, depth + 1, dc);
    // Run lengths are always copied: the start-flag step scans them in place. The copy is dead
    // after that step, so the length tables of all RLE nodes of a program share one region.
    const uint32_t lengths = add_seq(LOC_LDS, 0, runs, 4);
    transient.push_back(lengths);
    // The run values are copied first and run into the bitmap clear and the length copy without a
    // barrier (none of the three reads what another writes): the memory latencies of the two copies
    // overlap instead of adding up.
    const size_t before_values = steps.size();
    const uint32_t values = as_seq(vp, n);
    if (steps.size() == before_values + 1 && steps.back().kind == S_MATERIALIZE) steps.back().no_barrier = 1;
    const Bitmap bm = add_bitmap(n);
    add_step(S_BITMAP_CLEAR, NONE, bm.bits);
    steps.back().no_barrier = 1;  // the length copy touches neither the bitmap nor anything the clear reads
    add_step(S_MATERIALIZE, cp, lengths);
    add_step(S_BITMAP_STARTS, lengths, bm.bits, bm.rank, n, runs);  // fills the rank table too unless runs are very long
    add_step(S_BITMAP_RANK, 1, bm.bits, bm.rank, n, runs, ST_RLE_COUNTS);
    const uint32_t p = add_pipe(B_RLE, IS_DOUBLE, n);
    pipes[p].a[0] = bm.bits;
    pipes[p].a[1] = bm.rank;
    pipes[p].a[2] = values;
    pipes[p].a[3] = runs;
    desc = "RLE(" + dv + "," + dc + ")";
    return p;
  }

  template <bool IS_DOUBLE>
  uint32_t frequency_node(size_t off, uint32_t n, int depth, std::string& desc) {
    const uint32_t data_off = IS_DOUBLE ? FREQ_D_DATA : FREQ_I_DATA;
    const uint32_t exc_off = w(off + (IS_DOUBLE ? FREQ_D_EXC_OFFSET : FREQ_I_EXC_OFFSET));
    const uint8_t next = b(off + (IS_DOUBLE ? FREQ_D_NEXT : FREQ_I_NEXT));
    need(off, data_off, "Frequency header");
    needed += data_off;
    const size_t data = off + data_off;
    const Bitmap bm = add_bitmap(n);
    const uint32_t card = roaring_steps(data, bm, n, true);
    uint32_t exceptions = NONE;
    std::string de = "-";
    if (card > 0) {
      const uint32_t ep = IS_DOUBLE ? dbl_node(data + exc_off, next, card, depth + 1, de)
                                    : int_node(data + exc_off, next, card, false, depth + 1, de);
      exceptions = as_seq(ep);
    }
    const uint32_t p = add_pipe(B_FREQ, IS_DOUBLE, n);
    Pipe& pp = pipes[p];
    pp.a[0] = rd32(cs.host + off);
    pp.a[1] = IS_DOUBLE ? rd32(cs.host + off + 4) : 0;
    pp.a[2] = bm.bits;
    pp.a[3] = bm.rank;
    pp.a[4] = exceptions;
    pp.a[5] = card;
    desc = "FREQUENCY(" + de + ")";
    return p;
  }

  uint32_t int_node(size_t off, uint8_t scheme, uint32_t n, bool nullmap_allzeros, int depth,
                    std::string& desc) {
    if (depth > kMaxDepth) throw PlanError(ST_UNSUPPORTED, "cascade deeper than supported");
    max_depth = std::max<uint32_t>(max_depth, (uint32_t)depth + 1);
    switch (scheme) {
      case I_UNCOMPRESSED: {
        need(off, (size_t)n * 4, "UNCOMPRESSED data");
        needed += (uint64_t)n * 4;
        const uint32_t p = add_pipe(B_SEQ, false, n);
        pipes[p].a[0] = add_seq(LOC_SRC, (uint32_t)off, n, 4);
        desc = "UNCOMPRESSED";
        return p;
      }
      case I_ONE_VALUE: {
        const uint32_t v = w(off);
        needed += 4;
        const uint32_t p = add_pipe(B_ONE, false, n);
        pipes[p].a[0] = v;
        pipes[p].a[1] = 0;
        desc = "ONE_VALUE";
        return p;
      }
      case I_BP: return bp_node(off, n, n, false, desc);
      case I_DICT: {
        const uint8_t codes_scheme = b(off + DICT_CODES_SCHEME);
        const uint32_t codes_off = w(off + DICT_CODES_OFFSET);
        const size_t data = off + DICT_DATA;
        need(data, codes_off, "dictionary");
        const uint32_t dict_count = codes_off / 4;
        needed += DICT_DATA + (uint64_t)dict_count * 4;
        std::string dc;
        uint32_t child = int_node(data + codes_off, codes_scheme, n, false, depth + 1, dc);
        const uint32_t dict = dict_seq(data, dict_count, 4, n);
        desc = "DICT(" + dc + ")";
        return append_post(child, POST_DICT, dict);
      }
      case I_RLE: return rle_node<false>(off, n, depth, desc);
      case I_FREQUENCY: return frequency_node<false>(off, n, depth, desc);
      case I_FOR: {
        const uint32_t bias = w(off + FOR_BIAS);
        const uint8_t next = b(off + FOR_NEXT);
        needed += FOR_DATA;
        std::string dn;
        // FOR hands the chunk's nullmap to its child and skips the add when every row is null
        // (integer/FOR.cpp:53-59).
        uint32_t child = int_node(off + FOR_DATA, next, n, nullmap_allzeros, depth + 1, dn);
        desc = "FOR(" + dn + ")";
        if (nullmap_allzeros) return child;
        return append_post(child, POST_ADD, bias);
      }
      case I_DICTIONARY_8: return fixed_dict_node<uint8_t>(off, n, false, desc);
      case I_DICTIONARY_16: return fixed_dict_node<uint16_t>(off, n, false, desc);
      case I_PFOR: return pfor_node(off, n, false, desc);
      case I_PFOR_DELTA: return pfor_node(off, n, true, desc);
      case I_TRUNCATION_8:
      case I_TRUNCATION_16:
        // The reference cannot decode these either (integer/Truncation.hpp:88-97).
        throw PlanError(ST_UNSUPPORTED, "Truncation schemes have no decoder");
      default: corrupt("unknown integer scheme code " + std::to_string(scheme));
    }
  }

  uint32_t dbl_node(size_t off, uint8_t scheme, uint32_t n, int depth, std::string& desc) {
    if (depth > kMaxDepth) throw PlanError(ST_UNSUPPORTED, "cascade deeper than supported");
    max_depth = std::max<uint32_t>(max_depth, (uint32_t)depth + 1);
    switch (scheme) {
      case D_UNCOMPRESSED: {
        need(off, (size_t)n * 8, "UNCOMPRESSED data");
        needed += (uint64_t)n * 8;
        const uint32_t p = add_pipe(B_SEQ, true, n);
        pipes[p].a[0] = add_seq(LOC_SRC, (uint32_t)off, n, 8);
        desc = "UNCOMPRESSED";
        return p;
      }
      case D_ONE_VALUE: {
        need(off, 8, "ONE_VALUE");
        needed += 8;
        const uint32_t p = add_pipe(B_ONE, true, n);
        pipes[p].a[0] = rd32(cs.host + off);
        pipes[p].a[1] = rd32(cs.host + off + 4);
        desc = "ONE_VALUE";
        return p;
      }
      case D_DICT: {
        const uint8_t codes_scheme = b(off + DICT_CODES_SCHEME);
        const uint32_t codes_off = w(off + DICT_CODES_OFFSET);
        const size_t data = off + DICT_DATA;
        need(data, codes_off, "dictionary");
        const uint32_t dict_count = codes_off / 8;
        needed += DICT_DATA + (uint64_t)dict_count * 8;
        std::string dc;
        const uint32_t codes = int_node(data + codes_off, codes_scheme, n, false, depth + 1, dc);
        const uint32_t dict = dict_seq(data, dict_count, 8, n);
        const uint32_t p = add_pipe(B_DICTD, true, n);
        pipes[p].a[0] = codes;
        pipes[p].a[1] = dict;
        desc = "DICT(" + dc + ")";
        return p;
      }
      case D_RLE: return rle_node<true>(off, n, depth, desc);
      case D_FREQUENCY: return frequency_node<true>(off, n, depth, desc);
      case D_PSEUDODECIMAL: {
        need(off, DEC_DATA, "Pseudodecimal header");
        needed += DEC_DATA;
        const uint32_t converted = w(off + DEC_CONVERTED);
        if (converted > n) corrupt("Pseudodecimal converted_count above tuple count");
        const uint8_t ns = b(off + DEC_NUMBERS_SCHEME), es = b(off + DEC_EXPONENTS_SCHEME),
                      ps = b(off + DEC_PATCHES_SCHEME);
        const uint32_t e_off = w(off + DEC_EXPONENTS_OFFSET), p_off = w(off + DEC_PATCHES_OFFSET);
        const size_t data = off + DEC_DATA;
        std::string dn = "-", de, dp = "-";
        uint32_t digits = NONE;
        if (converted > 0) digits = int_node(data, ns, converted, false, depth + 1, dn);
        const uint32_t exps = int_node(data + e_off, es, n, false, depth + 1, de);
        uint32_t patches = NONE;
        if (n - converted > 0) {
          const uint32_t pp = dbl_node(data + p_off, ps, n - converted, depth + 1, dp);
          patches = as_seq(pp);
        }
        // The Roaring map of 4-row groups at exceptions_map_offset only steers the reference's
        // AVX strategy; exception rows are found from the exponent stream instead.
        const Bitmap bm = add_bitmap(n);
        add_step(S_BITMAP_CLEAR, NONE, bm.bits);
        add_step(S_BITMAP_EQ, exps, bm.bits, NONE, n, DEC_EXCEPTION_CODE);
        add_step(S_BITMAP_RANK, NONE, bm.bits, bm.rank, n, n - converted, ST_PSEUDO);
        const uint32_t p = add_pipe(B_PSEUDO, true, n);
        Pipe& q = pipes[p];
        q.a[0] = digits;
        q.a[1] = exps;
        q.a[2] = patches;
        q.a[3] = bm.bits;
        q.a[4] = bm.rank;
        q.a[5] = converted;
        desc = "PSEUDODECIMAL(" + dn + "," + de + "," + dp + ")";
        return p;
      }
      case D_DOUBLE_BP: return bp_node(off, n, n * 2, true, desc);
      case D_DICTIONARY_8: return fixed_dict_node<uint8_t>(off, n, true, desc);
      case D_DICTIONARY_16: return fixed_dict_node<uint16_t>(off, n, true, desc);
      default: corrupt("unknown double scheme code " + std::to_string(scheme));
    }
  }
};

uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
// Table sizes come from untrusted counts: always 64-bit products.
uint64_t seq_bytes(const Seq& s) { return align_up((uint64_t)s.count * s.esize, 16); }

}  // namespace

void build_program(const ColumnSource& src, BuiltProgram& out) {
  if (src.column_type != COL_INTEGER && src.column_type != COL_DOUBLE)
    throw PlanError(ST_UNSUPPORTED, "only integer and double columns are decoded on the GPU");
  // Blocks hold at most 65536 rows in column part files (btrblocks.hpp:24) and a few hundred
  // thousand in relation-level blocks; far larger counts only come from damaged headers.
  if (src.tuple_count > kMaxTupleCount)
    throw PlanError(ST_UNSUPPORTED, "tuple count above " + std::to_string(kMaxTupleCount) + " rows per block");
  if (src.bytes > 0xFFFFFFFFull) throw PlanError(ST_UNSUPPORTED, "compressed block above 4 GiB");
  Builder bld{src, {}, {}, {}, {}, {}, {}, {}, 0, 0};
  const uint32_t n = src.tuple_count;
  std::string desc;
  uint32_t root;
  if (src.column_type == COL_INTEGER)
    root = bld.int_node(src.payload_off, src.scheme, n, src.nullmap_type == NM_ALLZEROS, 0, desc);
  else
    root = bld.dbl_node(src.payload_off, src.scheme, n, 0, desc);
  const uint32_t dst = bld.add_seq(LOC_DST, 0, n, src.column_type == COL_DOUBLE ? 8 : 4);
  uint32_t kernel = src.column_type == COL_DOUBLE ? KERNEL_GENERIC_DBL : KERNEL_GENERIC_INT;
  uint32_t prog_base = 0;  // LDS byte offset of the program copy
  // A root that reads one full-length bit-packed stream behind at most one post-op (BP, FOR(BP),
  // DICT(BP)) goes to the stream kernels: its only preparation may be dictionary copies into
  // LDS, and its group index is built while decoding instead of in a separate pass.
  {
    const Pipe& rp = bld.pipes[root];
    // a byte-per-row nullmap is written by the stream kernels themselves, behind the root step
    // (a Roaring nullmap is expanded behind the root too, in the LDS the payload ring no longer
    // needs: its bitmap and rank table must fit there)
    const uint64_t nm_words = n / 64 + 1;
    const bool const_nullmap = !src.want_nullmap || src.nullmap_type == NM_ALLONES || src.nullmap_type == NM_ALLZEROS ||
                               ((src.nullmap_type == NM_REGULAR || src.nullmap_type == NM_FLIPPED) &&
                                align_up(nm_words * 8, 16) + nm_words * 4 <= kStreamRingBytes);
    bool eligible = src.column_type == COL_INTEGER && rp.base == B_BP && rp.n_post <= 1 &&
                    const_nullmap && bld.bp_steps.size() == 1 && n > 0;
    for (const Step& st : bld.steps)
      eligible = eligible && st.kind == S_MATERIALIZE && bld.pipes[st.a[0]].base == B_SEQ &&
                 bld.pipes[st.a[0]].n_post == 0 && bld.seqs[st.a[1]].esize == 4;
    if (eligible) {
      bld.bp_steps.clear();
      bld.add_step(S_STREAM_ROOT, root, dst);
      prog_base = kStreamProgBase;  // the payload ring occupies the start of LDS
      if (rp.n_post == 0)
        kernel = KERNEL_STREAM_NONE;
      else if (rp.post[0].type == POST_ADD)
        kernel = KERNEL_STREAM_ADD;
      else
        kernel = KERNEL_STREAM_DICT_LDS;  // corrected below once table placement is known
    } else if (src.column_type == COL_DOUBLE && rp.base == B_DICTD && const_nullmap && n > 0 &&
               bld.bp_steps.size() == 1 && bld.pipes[rp.a[0]].base == B_BP && bld.pipes[rp.a[0]].n_post == 0 &&
               bld.seqs[rp.a[1]].count > 0 &&
               std::all_of(bld.steps.begin(), bld.steps.end(), [&](const Step& st) {
                 return st.kind == S_MATERIALIZE && bld.pipes[st.a[0]].base == B_SEQ && bld.pipes[st.a[0]].n_post == 0;
               })) {
      // DICT(BP) doubles: codes through the ring, 8-byte dictionary gather
      bld.bp_steps.clear();
      bld.add_step(S_STREAM_ROOT, root, dst);
      prog_base = kStreamProgBase;
      kernel = KERNEL_STREAM_DICTD;
    } else if (src.column_type == COL_DOUBLE && rp.base == B_PSEUDO && rp.a[0] != NONE && const_nullmap &&
               n > 0 && n / 64 + 1 <= 4096 && bld.bp_steps.size() == 2 &&
               bld.pipes[rp.a[0]].base == B_BP && bld.pipes[rp.a[0]].n_post == 0 &&
               bld.pipes[rp.a[1]].base == B_BP && (rp.a[2] == NONE || bld.seqs[rp.a[2]].loc == LOC_SRC) &&
               bld.stream_exponents(rp.a[1], src.host) && bld.steps.size() == 3) {
      // PSEUDODECIMAL(BP, BP or DICT(BP) with a small dictionary of valid exponents) with raw (or
      // no) patches: both streams go through the LDS ring, the exponents first (exception flags
      // into the bitmap the pipe already owns), then the digits. The program keeps the clear and
      // rank steps; the kernel does the exponent pass in between.
      bld.bp_steps.clear();
      std::vector<Step> keep;
      for (const Step& st : bld.steps)
        if (st.kind != S_BITMAP_EQ) keep.push_back(st);
      bld.steps = keep;
      bld.add_step(S_STREAM_ROOT, root, dst);  // the powers-of-ten table sits at kStreamFracBase
      prog_base = kStreamProgBase;
      kernel = KERNEL_STREAM_PSEUDO;
    } else {
      // a root that is just a decoded PFOR table: decode straight into the column
      bool direct = false;
      if (rp.base == B_SEQ && rp.n_post == 0)
        for (Step& st : bld.steps)
          if (st.kind == S_PFOR_DECODE && st.a[4] == rp.a[0]) {
            bld.seqs[st.a[4]].count = 0;
            st.a[4] = dst;
            direct = true;
          }
      if (!direct) bld.add_step(S_MATERIALIZE, root, dst);
    }
  }
  if (kernel_is_generic(kernel))
    for (const Step& st : bld.steps)
      if (st.kind == S_PFOR_DECODE) kernel = src.column_type == COL_DOUBLE ? KERNEL_PFOR_DBL : KERNEL_PFOR_INT;
  // LDS scratch for the header-chain walkers: index step s runs on wave s % 4, so four slots
  // cover every program; the streamed root's walker starts after the index steps and reuses slot 0.
  uint32_t walk_scratch[4] = {NONE, NONE, NONE, NONE};
  {
    auto scratch_slot = [&](size_t k) {
      if (walk_scratch[k] == NONE) walk_scratch[k] = bld.add_seq(LOC_LDS, 0, kWalkScratchBytes / 16, 16, true);
      return walk_scratch[k];
    };
    for (size_t k = 0; k < bld.bp_steps.size(); k++)
      if (bld.pipes[bld.bp_steps[k].a[0]].a[5] > kWalkScratchMinGroups) bld.bp_steps[k].a[1] = scratch_slot(k % 4);
    if (!kernel_is_generic(kernel)) bld.steps.back().a[3] = scratch_slot(0);
  }
  uint32_t ring_bits = NONE, ring_rank = NONE;
  if (src.want_nullmap) {
    if (src.nullmap_type == NM_REGULAR || src.nullmap_type == NM_FLIPPED) {
      const Builder::Bitmap bm = bld.add_bitmap(n);
      bld.roaring_steps(src.nullmap_off, bm, n, true);
      bld.add_step(S_NULLMAP, src.nullmap_type, bm.bits, NONE, n);
      if (!kernel_is_generic(kernel)) {
        ring_bits = bm.bits;  // placed in the payload ring, which is idle behind the root step
        ring_rank = bm.rank;
        bld.seqs[bm.bits].must_lds = bld.seqs[bm.rank].must_lds = 3;
      }
    } else if (src.nullmap_type == NM_ALLONES || src.nullmap_type == NM_ALLZEROS) {
      bld.add_step(S_NULLMAP, src.nullmap_type, NONE, NONE, n);
    } else {
      throw PlanError(ST_CORRUPT, "unknown nullmap type");
    }
  }

  // ---- lay the program out, then place tables: mandatory LDS tables first, optional ones in
  // LDS while the soft cap holds, the rest in the chunk's scratch slice.
  const size_t n_steps = bld.bp_steps.size() + bld.steps.size();
  const size_t prog_bytes = sizeof(ProgramHeader) + n_steps * sizeof(Step) + bld.pipes.size() * sizeof(Pipe) +
                            bld.seqs.size() * sizeof(Seq) + bld.consts.size() * 4;
  if (prog_bytes > kLdsCapBytes) throw PlanError(ST_UNSUPPORTED, "decode program of this chunk exceeds the LDS budget");
  const uint32_t table_base = prog_base + (uint32_t)align_up(prog_bytes, 16) + 64;  // + reduction scratch
  if (ring_bits != NONE) {
    // offsets are relative to the table base: these two wrap around to LDS addresses 0.. (the ring)
    bld.seqs[ring_bits].off = 0u - table_base;
    bld.seqs[ring_rank].off = (uint32_t)align_up((uint64_t)bld.seqs[ring_bits].count * 8, 16) - table_base;
  }
  uint64_t lds = 0, scratch = 0;
  auto place_tables = [&]() {
    lds = scratch = 0;
    for (auto& s : bld.seqs)
      if (s.loc == LOC_LDS && s.must_lds && s.must_lds != 3) {
        s.off = (uint32_t)lds;
        lds += seq_bytes(s);
      }
    if (table_base + lds > kLdsCapBytes && kernel_is_generic(kernel)) {
      // very large blocks: give up the walkers' jump tables (they only speed the index steps up)
      for (uint32_t k = 0; k < 4; k++)
        if (walk_scratch[k] != NONE) {
          lds -= seq_bytes(bld.seqs[walk_scratch[k]]);
          bld.seqs[walk_scratch[k]].count = 0;
        }
      for (Step& st : bld.bp_steps) st.a[1] = NONE;
      auto place_mandatory = [&]() {
        lds = 0;
        for (auto& s : bld.seqs)
          if (s.loc == LOC_LDS && s.must_lds && s.must_lds != 3) {
            s.off = (uint32_t)lds;
            lds += seq_bytes(s);
          }
      };
      place_mandatory();
      if (table_base + lds > kLdsCapBytes) {
        // still too large: PFOR steps patch in place and do without their exception-merge tables
        for (Step& st : bld.steps)
          if (st.kind == S_PFOR_DECODE) {
            st.a[3] |= PFOR_NO_MERGE;
            bld.seqs[st.a[5]].count = kPforWorkBaseBytes / 16;
          }
        place_mandatory();
      }
    }
    {
      // one shared region for the transient tables, sized for the largest
      uint64_t shared = 0;
      for (uint32_t id : bld.transient) shared = std::max<uint64_t>(shared, seq_bytes(bld.seqs[id]));
      const uint32_t cap = kernel_is_generic(kernel) ? kLdsSoftCapGeneric : kLdsSoftCapStream;
      if (shared && table_base + lds + shared <= cap) {
        for (uint32_t id : bld.transient) {
          bld.seqs[id].off = (uint32_t)lds;
          bld.seqs[id].must_lds = 2;  // placed
        }
        lds += shared;
      }
    }
    for (auto& s : bld.seqs)
      if (s.loc == LOC_LDS && !s.must_lds) {
        const uint64_t bytes = seq_bytes(s);
        if (table_base + lds + bytes <= (kernel_is_generic(kernel) ? kLdsSoftCapGeneric : kLdsSoftCapStream)) {
          s.off = (uint32_t)lds;
          lds += bytes;
        } else {
          s.loc = LOC_SCRATCH;
          s.off = (uint32_t)scratch;
          scratch += bytes;
        }
      }
  };
  const std::vector<Seq> unplaced = bld.seqs;
  place_tables();
  if (kernel_is_generic(kernel) && !bld.transient.empty() && bld.seqs[bld.transient[0]].must_lds == 2) {
    // The region shared by the run-length tables is free once the last of them has been scanned.
    // A value table that found no room in LDS and is read by nothing but the root step can be
    // filled right before the root step instead of where its node was parsed, and then lives in
    // that region: the root step gathers from LDS instead of the scratch slice.
    const uint32_t region_off = bld.seqs[bld.transient[0]].off;
    uint64_t region_bytes = 0;
    for (uint32_t id : bld.transient) region_bytes = std::max<uint64_t>(region_bytes, seq_bytes(bld.seqs[id]));
    auto is_transient = [&](uint32_t id) { return std::find(bld.transient.begin(), bld.transient.end(), id) != bld.transient.end(); };
    std::function<void(uint32_t, std::vector<uint32_t>&)> reads = [&](uint32_t pipe, std::vector<uint32_t>& out) {
      if (pipe == NONE) return;
      const Pipe& pp = bld.pipes[pipe];
      auto seq = [&](uint32_t id) {
        if (id != NONE) out.push_back(id);
      };
      switch (pp.base) {
        case B_SEQ: seq(pp.a[0]); break;
        case B_BP:
        case B_DBP: seq(pp.a[2]), seq(pp.a[4]); break;
        case B_RLE: seq(pp.a[0]), seq(pp.a[1]), seq(pp.a[2]); break;
        case B_FREQ: seq(pp.a[2]), seq(pp.a[3]), seq(pp.a[4]); break;
        case B_DICTD: reads(pp.a[0], out), seq(pp.a[1]); break;
        case B_PSEUDO: reads(pp.a[0], out), reads(pp.a[1], out), seq(pp.a[2]), seq(pp.a[3]), seq(pp.a[4]); break;
        default: break;
      }
      for (int k = 0; k < pp.n_post; k++)
        if (pp.post[k].type == POST_DICT) seq(pp.post[k].arg);
    };
    size_t root_idx = bld.steps.size();
    for (size_t k = 0; k < bld.steps.size(); k++)
      if (bld.steps[k].kind == S_MATERIALIZE && bld.steps[k].a[1] == dst) root_idx = k;
    uint32_t best = NONE;
    size_t best_step = 0;
    for (uint32_t id = 0; id < bld.seqs.size() && root_idx < bld.steps.size(); id++) {
      const Seq& sq = bld.seqs[id];
      const uint64_t bytes = seq_bytes(sq);
      if (sq.loc != LOC_SCRATCH || bytes == 0 || bytes > region_bytes) continue;
      if (best != NONE && bytes <= seq_bytes(bld.seqs[best])) continue;
      size_t producer = bld.steps.size(), producers = 0;
      bool movable = true;
      for (size_t k = 0; k < bld.steps.size(); k++) {
        const Step& st = bld.steps[k];
        if (st.kind == S_MATERIALIZE && st.a[1] == id) {
          producer = k;
          producers++;
          continue;
        }
        if (st.kind == S_PFOR_DECODE && st.a[4] == id) movable = false;
        if (k == root_idx) continue;
        std::vector<uint32_t> used;
        if (st.kind == S_MATERIALIZE || st.kind == S_BITMAP_EQ) reads(st.a[0], used);
        if (st.kind == S_BITMAP_STARTS) used.push_back(st.a[0]);
        if (std::find(used.begin(), used.end(), id) != used.end()) movable = false;
      }
      if (!movable || producers != 1 || producer > root_idx) continue;
      std::vector<uint32_t> inputs;
      reads(bld.steps[producer].a[0], inputs);
      if (std::any_of(inputs.begin(), inputs.end(), is_transient)) continue;
      best = id;
      best_step = producer;
    }
    if (best != NONE) {
      Step moved = bld.steps[best_step];
      moved.no_barrier = 0;  // the root step right behind it reads the table
      // a step allowed to run into the moved one without a barrier gets another successor
      if (best_step > 0 && best_step + 1 != root_idx) bld.steps[best_step - 1].no_barrier = 0;
      bld.steps.erase(bld.steps.begin() + best_step);
      bld.steps.insert(bld.steps.begin() + (root_idx - 1), moved);  // root_idx shifted down by the erase
      bld.seqs[best].loc = LOC_LDS;
      bld.seqs[best].off = region_off;
      scratch = 0;  // the scratch slice ends behind the last table left in it
      for (const Seq& sq : bld.seqs)
        if (sq.loc == LOC_SCRATCH) scratch = std::max<uint64_t>(scratch, sq.off + seq_bytes(sq));
    }
  }
  size_t stream_root = bld.steps.size() - 1;  // a constant-nullmap step may follow the streamed root
  while (stream_root > 0 && bld.steps[stream_root].kind != S_STREAM_ROOT) stream_root--;
  if (!kernel_is_generic(kernel) && kernel != KERNEL_STREAM_PSEUDO && bld.steps[stream_root].a[3] != NONE) {
    // A stream kernel runs at most four 512-thread workgroups per CU, fewer when LDS runs out. The
    // walker's jump table only speeds up streams whose group lengths vary: when all groups of this
    // stream are equally long, give it up if that lets one more workgroup share the CU.
    // (the double-dictionary kernel runs 1024-thread workgroups: two per CU at most)
    const uint64_t most = kernel == KERNEL_STREAM_DICTD ? 2 : 4;
    auto per_cu = [&](uint64_t bytes) { return std::min<uint64_t>(most, kLdsCapBytes / bytes); };
    const uint64_t with = table_base + lds, without = with - align_up(kWalkScratchBytes, 16);
    const Pipe& bp = kernel == KERNEL_STREAM_DICTD ? bld.pipes[bld.pipes[root].a[0]] : bld.pipes[root];
    auto equally_long = [&]() {
      const uint8_t* words = src.host + bp.a[0];
      const uint64_t W = bp.a[1];
      uint64_t pos = 1, first_len = 0;  // word 0 is the value count
      for (uint32_t g = 0; g + 1 < bp.a[5]; g++) {  // the length of the last group ends no step
        if (pos >= W) return false;
        const uint8_t* h = words + 4 * pos;
        const uint64_t len = 1 + h[0] + h[1] + h[2] + h[3];
        if (g == 0) first_len = len;
        if (len != first_len) return false;
        pos += len;
      }
      return true;
    };
    if (per_cu(without) > per_cu(with) && equally_long()) {
      const uint32_t jump = bld.steps[stream_root].a[3];
      bld.seqs = unplaced;
      bld.seqs[jump].count = 0;
      bld.steps[stream_root].a[3] = NONE;
      place_tables();
    }
  }
  if (table_base + lds > kLdsCapBytes)
    throw PlanError(ST_UNSUPPORTED, "decode tables of this chunk exceed the LDS budget (" +
                                        std::to_string(table_base + lds) + " bytes)");
  if (scratch > 0xFFFFFFFFull) throw PlanError(ST_UNSUPPORTED, "scratch need above 4 GiB");

  if (kernel == KERNEL_STREAM_DICTD && bld.seqs[bld.pipes[root].a[1]].loc == LOC_SCRATCH) {
    // the LDS copy did not fit: gather from the dictionary inside the compressed bytes instead
    for (size_t k = 0; k < bld.steps.size(); k++)
      if (bld.steps[k].kind == S_MATERIALIZE && bld.steps[k].a[1] == bld.pipes[root].a[1]) {
        bld.pipes[root].a[1] = bld.pipes[bld.steps[k].a[0]].a[0];
        bld.steps.erase(bld.steps.begin() + k);
        scratch = 0;
        break;
      }
  }
  if (kernel == KERNEL_STREAM_DICT_LDS) {
    const Seq& dict = bld.seqs[bld.pipes[root].post[0].arg];
    if (dict.count == 0) throw PlanError(ST_CORRUPT, "empty dictionary");
    if (dict.loc == LOC_SCRATCH) {
      // the LDS copy did not fit: gather from the dictionary inside the compressed bytes instead
      for (size_t k = 0; k < bld.steps.size(); k++)
        if (bld.steps[k].kind == S_MATERIALIZE && bld.steps[k].a[1] == bld.pipes[root].post[0].arg) {
          bld.pipes[root].post[0].arg = bld.pipes[bld.steps[k].a[0]].a[0];
          bld.steps.erase(bld.steps.begin() + k);
          scratch = 0;
          break;
        }
    }
    if (bld.seqs[bld.pipes[root].post[0].arg].loc != LOC_LDS) kernel = KERNEL_STREAM_DICT_GLB;
  }
  const size_t n_steps_final = bld.bp_steps.size() + bld.steps.size();
  const size_t consts_off = sizeof(ProgramHeader) + n_steps_final * sizeof(Step) + bld.pipes.size() * sizeof(Pipe) +
                            bld.seqs.size() * sizeof(Seq);
  const size_t prog_bytes_final = consts_off + bld.consts.size() * 4;
  out.program.assign(prog_bytes_final, 0);
  uint8_t* p = out.program.data();
  ProgramHeader hdr{};
  hdr.n_steps = (uint32_t)n_steps_final;
  hdr.n_bp_steps = (uint32_t)bld.bp_steps.size();
  hdr.n_pipes = (uint32_t)bld.pipes.size();
  hdr.n_seqs = (uint32_t)bld.seqs.size();
  hdr.table_base = table_base;
  hdr.consts_off = (uint32_t)consts_off;
  hdr.magic = kProgramMagic;
  hdr.prog_bytes = (uint32_t)prog_bytes_final;
  if (prog_bytes_final > kMaxProgramBytes)
    throw PlanError(ST_UNSUPPORTED, "decode program of this chunk above " + std::to_string(kMaxProgramBytes) + " bytes");
  memcpy(p, &hdr, sizeof(hdr));
  p += sizeof(hdr);
  if (!bld.bp_steps.empty()) memcpy(p, bld.bp_steps.data(), bld.bp_steps.size() * sizeof(Step));
  p += bld.bp_steps.size() * sizeof(Step);
  if (!bld.steps.empty()) memcpy(p, bld.steps.data(), bld.steps.size() * sizeof(Step));
  p += bld.steps.size() * sizeof(Step);
  memcpy(p, bld.pipes.data(), bld.pipes.size() * sizeof(Pipe));
  p += bld.pipes.size() * sizeof(Pipe);
  memcpy(p, bld.seqs.data(), bld.seqs.size() * sizeof(Seq));
  p += bld.seqs.size() * sizeof(Seq);
  if (!bld.consts.empty()) memcpy(p, bld.consts.data(), bld.consts.size() * 4);
  out.kernel = !kernel_is_generic(kernel) && src.want_nullmap ? kernel + KERNEL_STREAM_NULLMAP : kernel;
  out.lds_bytes = table_base + (uint32_t)lds;
  out.scratch_bytes = (uint32_t)scratch;
  out.needed_bytes = bld.needed;
  out.depth = bld.max_depth;
  out.description = desc;
}

}  // namespace btrplan
