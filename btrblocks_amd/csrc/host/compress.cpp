// CPU compressor for BtrBlocks integer/double column chunks (see compress.h).
//
// Each scheme writer follows the on-wire layout of the reference's compressor:
//   RLE            btrblocks/scheme/templated/RLE.hpp:20-85
//   DICT           btrblocks/scheme/templated/DynamicDictionary.hpp:29-75
//   FREQUENCY      btrblocks/scheme/templated/Frequency.hpp:50-97
//   PSEUDODECIMAL  btrblocks/scheme/double/Pseudodecimal.cpp:56-184
//   FOR            btrblocks/scheme/integer/FOR.cpp:18-44
//   BP / PFOR      btrblocks/scheme/integer/PBP.cpp:18-35, 126-143
//   DOUBLE_BP      btrblocks/scheme/double/DoubleBP.cpp:16-39
//   ONE_VALUE      btrblocks/scheme/integer/OneValue.cpp:20-28
//   DICTIONARY_8/16 btrblocks/scheme/templated/FixedDictionary.hpp:25-54
//   nullmap        btrblocks/extern/RoaringBitmap.cpp:117-159
//   envelope       btrblocks/compression/Datablock.cpp:66-148, 192-338
#include "compress.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <stdexcept>

#include "../btr_format.h"
#include "codecs.h"

namespace btrhost {
using namespace btrfmt;

const Spec& Spec::kid(size_t i) const {
  static const Spec kAuto;
  return i < kids.size() ? kids[i] : kAuto;
}

// ------------------------------------------------------------------------------------------
// spec parser
// ------------------------------------------------------------------------------------------
namespace {
struct NameCode {
  const char* name;
  int icode, dcode;
};
const NameCode kNames[] = {
    {"auto", SCHEME_AUTO, SCHEME_AUTO},
    {"uncompressed", I_UNCOMPRESSED, D_UNCOMPRESSED},
    {"onevalue", I_ONE_VALUE, D_ONE_VALUE},
    {"dict", I_DICT, D_DICT},
    {"rle", I_RLE, D_RLE},
    {"pfor", I_PFOR, -1},
    {"bp", I_BP, -1},
    {"frequency", I_FREQUENCY, D_FREQUENCY},
    {"for", I_FOR, -1},
    {"pfor_delta", I_PFOR_DELTA, -1},
    {"dict8", I_DICTIONARY_8, D_DICTIONARY_8},
    {"dict16", I_DICTIONARY_16, D_DICTIONARY_16},
    {"pseudodecimal", -1, D_PSEUDODECIMAL},
    {"doublebp", -1, D_DOUBLE_BP},
};

// Types of a scheme's children, in child order: 'i' integer stream, 'd' double stream,
// 's' same type as the parent.
const char* child_types(bool is_double, uint8_t code) {
  if (is_double) {
    switch (code) {
      case D_DICT: return "i";
      case D_RLE: return "si";
      case D_FREQUENCY: return "s";
      case D_PSEUDODECIMAL: return "iid";
      default: return "";
    }
  }
  switch (code) {
    case I_DICT: return "i";
    case I_RLE: return "si";
    case I_FREQUENCY: return "s";
    case I_FOR: return "s";
    default: return "";
  }
}

struct Parser {
  const char* p;
  std::string err;
  void ws() {
    while (*p && isspace((unsigned char)*p)) p++;
  }
  bool node(bool is_double, Spec& out) {
    ws();
    std::string name;
    while (*p && (isalnum((unsigned char)*p) || *p == '_')) name.push_back((char)tolower(*p++));
    int code = -2;
    for (auto& nc : kNames)
      if (name == nc.name) code = is_double ? nc.dcode : nc.icode;
    if (code == -2) return err = "unknown scheme name '" + name + "'", false;
    if (code == -1)
      return err = "scheme '" + name + "' does not exist for this column type", false;
    out.scheme = (uint8_t)code;
    out.kids.clear();
    ws();
    if (*p != '(') return true;
    p++;
    const char* types = child_types(is_double, out.scheme);
    for (size_t k = 0;; k++) {
      if (!types[k]) return err = "too many children for '" + name + "'", false;
      const bool kid_double = types[k] == 'd' || (types[k] == 's' && is_double);
      Spec kid;
      if (!node(kid_double, kid)) return false;
      out.kids.push_back(kid);
      ws();
      if (*p == ',') {
        p++;
        continue;
      }
      if (*p == ')') {
        p++;
        return true;
      }
      return err = "expected ',' or ')'", false;
    }
  }
};
}  // namespace

bool parse_spec(const char* text, bool is_double, Spec& out, std::string& err) {
  Parser ps{text ? text : "auto", {}};
  if (!ps.node(is_double, out)) return err = ps.err, false;
  ps.ws();
  if (*ps.p) return err = "trailing characters in scheme spec", false;
  return true;
}

// ------------------------------------------------------------------------------------------
// writer: a growing byte buffer whose `origin` is assumed to sit on a 16-byte boundary
// ------------------------------------------------------------------------------------------
namespace {
struct Writer {
  std::vector<uint8_t>& buf;
  size_t origin;
  size_t pos() const { return buf.size(); }
  size_t reserve(size_t n) {
    size_t at = buf.size();
    buf.resize(at + n, 0);
    return at;
  }
  void put(const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    buf.insert(buf.end(), b, b + n);
  }
  uint8_t* at(size_t off) { return buf.data() + off; }
  // bytes needed so that the next byte lands on a multiple of `a` relative to origin
  size_t pad_to(size_t a) const { return (a - ((buf.size() - origin) % a)) % a; }
};

struct Ctx {
  const Options& opt;
  bool estimating;  // inside a ratio estimate: nested picks compress their whole input
};

template <typename T>
struct Stats {
  std::vector<T> distinct;       // ascending, duplicate-free (by ==)
  std::vector<uint32_t> counts;  // occurrences per distinct value
  T min{}, max{};
  uint32_t n = 0, null_count = 0;
  double avg_run_length = 0;
};

// Total order behind the distinct-value tables: numbers by value (so -0.0 and 0.0 share an entry,
// like the reference's std::map), NaNs after every number and among themselves by bit pattern.
// The reference's map is undefined for NaN; here a NaN row keeps its exact bits.
template <typename T>
inline bool value_less(T a, T b) {
  if constexpr (sizeof(T) == 8) {
    const bool an = a != a, bn = b != b;
    if (an || bn) {
      if (!(an && bn)) return bn;
      uint64_t x, y;
      memcpy(&x, &a, 8);
      memcpy(&y, &b, 8);
      return x < y;
    }
  }
  return a < b;
}
template <typename T>
inline bool value_same(T a, T b) {
  return !value_less(a, b) && !value_less(b, a);
}

// Mirrors NumberStats<T>::generateStats (btrblocks/stats/NumberStats.hpp:78-130): values at null
// rows take part in distinct/min/max; a run only breaks at a non-null row with a new value.
template <typename T>
Stats<T> make_stats(const T* src, const uint8_t* nm, uint32_t n) {
  Stats<T> s;
  s.n = n;
  if (n == 0) return s;
  std::vector<T> sorted(src, src + n);
  std::sort(sorted.begin(), sorted.end(), value_less<T>);
  for (uint32_t i = 0; i < n;) {
    uint32_t j = i;
    while (j < n && value_same(sorted[j], sorted[i])) j++;
    s.distinct.push_back(sorted[i]);
    s.counts.push_back(j - i);
    i = j;
  }
  s.min = s.distinct.front();
  s.max = s.distinct.back();
  T last = src[0];
  uint32_t runs = 1;
  for (uint32_t i = 0; i < n; i++) {
    const bool valid = nm == nullptr || nm[i];
    if (src[i] != last && valid) {
      last = src[i];
      runs++;
    }
    if (!valid) s.null_count++;
  }
  s.avg_run_length = (double)(uint32_t)((double)n / (double)runs);  // u32 field in the reference
  return s;
}

uint32_t bits_needed_s32(int32_t v) {
  if (v < 0) return 32;
  if (v == 0) return 1;
  return 32u - (uint32_t)__builtin_clz((uint32_t)v);
}

// Deterministic stand-in for NumberStats::samples (NumberStats.hpp:34-76): `count` slices of
// `len` rows, each from the middle of its partition.
template <typename T>
void take_sample(const T* src, const uint8_t* nm, uint32_t n, const Options& o, std::vector<T>& sv,
                 std::vector<uint8_t>& sn) {
  const uint32_t cnt = o.sample_count, len = o.sample_size;
  if (n <= cnt * len) {
    sv.assign(src, src + n);
    if (nm)
      sn.assign(nm, nm + n);
    else
      sn.assign(n, 1);
    return;
  }
  const uint32_t part = n / cnt;
  for (uint32_t s = 0; s < cnt; s++) {
    const uint32_t begin = s * part + (part - len) / 2;
    sv.insert(sv.end(), src + begin, src + begin + len);
    if (nm)
      sn.insert(sn.end(), nm + begin, nm + begin + len);
    else
      sn.insert(sn.end(), len, 1);
  }
}

uint8_t compress_int(Writer& w, const Ctx& cx, const int32_t* src, const uint8_t* nm, uint32_t n,
                     int level, const Spec& spec, bool top);
uint8_t compress_dbl(Writer& w, const Ctx& cx, const double* src, const uint8_t* nm, uint32_t n,
                     int level, const Spec& spec, bool top);

// ---- leaf payloads -----------------------------------------------------------------------
void emit_fastpfor(Writer& w, const uint32_t* vals, size_t n, bool pfor) {
  const size_t hdr = w.reserve(BP_DATA);
  // The reference aligns the payload relative to the absolute address of `data`
  // (PBP.cpp:24-28, 132-136); chunk origins are 16-byte aligned, so offsets reproduce it.
  const size_t pad = w.pad_to(pfor ? 16 : 4);
  w.reserve(pad);
  std::vector<uint32_t> words;
  if (pfor)
    pfor_encode(vals, n, words, ((w.pos() - w.origin) / 4) % 4);
  else
    fbp_encode(vals, n, words);
  w.put(words.data(), words.size() * 4);
  wr32(w.at(hdr + BP_WORDS), (uint32_t)words.size());
  *w.at(hdr + BP_PADDING) = (uint8_t)pad;
  // The reference reports sizeof(XPBPStructure) = 8 bytes of header although `data` starts at
  // offset 5, plus 16 bytes of slack for PFOR (PBP.cpp:34, 142); parents continue writing after
  // the reported size, so pad the stream to it.
  const size_t reported = 8 + words.size() * 4 + (pfor ? 16 : 0);
  const size_t written = w.pos() - hdr;
  if (written < reported) w.reserve(reported - written);
}

template <typename T>
void emit_rle(Writer& w, const Ctx& cx, const T* src, const uint8_t* nm, uint32_t n, int level,
              const Spec& spec) {
  std::vector<T> values;
  std::vector<int32_t> counts;
  T last = src[0];
  int32_t count = 1;
  for (uint32_t i = 1; i < n; i++) {
    if (value_same(src[i], last) || (nm != nullptr && !nm[i])) {
      count++;
    } else {
      counts.push_back(count);
      values.push_back(last);
      last = src[i];
      count = 1;
    }
  }
  counts.push_back(count);
  values.push_back(last);
  const size_t hdr = w.reserve(RLE_DATA);
  const size_t data = w.pos();
  wr32(w.at(hdr + RLE_RUNS), (uint32_t)counts.size());
  uint8_t vcode;
  if constexpr (sizeof(T) == 4)
    vcode = compress_int(w, cx, values.data(), nullptr, (uint32_t)values.size(), level - 1,
                         spec.kid(0), false);
  else
    vcode = compress_dbl(w, cx, values.data(), nullptr, (uint32_t)values.size(), level - 1,
                         spec.kid(0), false);
  wr32(w.at(hdr + RLE_COUNTS_OFFSET), (uint32_t)(w.pos() - data));
  const size_t cstart = w.pos();
  const uint8_t ccode = compress_int(w, cx, counts.data(), nullptr, (uint32_t)counts.size(),
                                     level - 1, spec.kid(1), false);
  // The reference advances its write pointer by the counts size twice (RLE.hpp:75-77); the
  // second span is dead bytes that still count towards the stored size.
  w.reserve(w.pos() - cstart);
  *w.at(hdr + RLE_VALUES_SCHEME) = vcode;
  *w.at(hdr + RLE_COUNTS_SCHEME) = ccode;
  // sizeof(RLEStructure) is 12 although data starts at 10: nothing to add, the reference
  // computes the size from pointers.
}

template <typename T>
void emit_dict(Writer& w, const Ctx& cx, const T* src, uint32_t n, int level, const Spec& spec,
               const Stats<T>& st) {
  const size_t hdr = w.reserve(DICT_DATA);
  const size_t data = w.pos();
  w.put(st.distinct.data(), st.distinct.size() * sizeof(T));
  std::vector<int32_t> codes(n);
  for (uint32_t i = 0; i < n; i++) {
    auto it = std::lower_bound(st.distinct.begin(), st.distinct.end(), src[i], value_less<T>);
    codes[i] = (int32_t)(it - st.distinct.begin());
  }
  wr32(w.at(hdr + DICT_CODES_OFFSET), (uint32_t)(w.pos() - data));
  // Codes go to BP unless the caller names something else (DynamicDictionary.hpp:62-67).
  Spec bp;
  bp.scheme = I_BP;
  const Spec& kid = spec.kids.empty() ? bp : spec.kid(0);
  const uint8_t ccode = compress_int(w, cx, codes.data(), nullptr, n, level - 1, kid, false);
  *w.at(hdr + DICT_CODES_SCHEME) = ccode;
}

template <typename T>
void emit_frequency(Writer& w, const Ctx& cx, const T* src, const uint8_t* nm, uint32_t n,
                    int level, const Spec& spec, const Stats<T>& st) {
  constexpr bool is_int = sizeof(T) == 4;
  const uint32_t data_off = is_int ? FREQ_I_DATA : FREQ_D_DATA;
  const size_t hdr = w.reserve(data_off);
  T top;
  if ((double)st.null_count * 100.0 / (double)n >= 90) {
    top = (T)0;  // NULL_CODE
  } else {
    top = st.distinct[0];
    uint32_t best = st.counts[0];
    for (size_t i = 0; i < st.distinct.size(); i++)
      if (st.counts[i] > best) best = st.counts[i], top = st.distinct[i];
  }
  std::vector<T> exceptions;
  std::vector<uint32_t> rows;
  for (uint32_t i = 0; i < n; i++)
    if (!value_same(src[i], top) && (nm == nullptr || nm[i])) exceptions.push_back(src[i]), rows.push_back(i);
  memcpy(w.at(hdr), &top, sizeof(T));
  const size_t data = w.pos();
  roaring_serialize(rows.data(), rows.size(), w.buf);
  wr32(w.at(hdr + (is_int ? FREQ_I_EXC_OFFSET : FREQ_D_EXC_OFFSET)), (uint32_t)(w.pos() - data));
  uint8_t code;
  if constexpr (is_int)
    code = compress_int(w, cx, exceptions.data(), nullptr, (uint32_t)exceptions.size(), level - 1,
                        spec.kid(0), false);
  else
    code = compress_dbl(w, cx, exceptions.data(), nullptr, (uint32_t)exceptions.size(), level - 1,
                        spec.kid(0), false);
  *w.at(hdr + (is_int ? FREQ_I_NEXT : FREQ_D_NEXT)) = code;
}

template <typename T, typename Code>
void emit_fixed_dict(Writer& w, const T* src, uint32_t n, const Stats<T>& st) {
  const uint32_t slots = sizeof(T) == 4 ? FDICT_I_SLOTS : FDICT_D_SLOTS;
  const size_t hdr = w.reserve(slots);
  w.put(st.distinct.data(), st.distinct.size() * sizeof(T));
  wr32(w.at(hdr + FDICT_CODES_OFFSET), (uint32_t)(w.pos() - hdr));
  for (uint32_t i = 0; i < n; i++) {
    auto it = std::lower_bound(st.distinct.begin(), st.distinct.end(), src[i], value_less<T>);
    Code c = (Code)(it - st.distinct.begin());
    w.put(&c, sizeof(Code));
  }
}

const double kFractions[23] = {
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

// Returns false when more than half the rows are patches (Pseudodecimal.cpp:114-119).
bool emit_pseudodecimal(Writer& w, const Ctx& cx, const double* src, uint32_t n, int level,
                        const Spec& spec) {
  std::vector<int32_t> numbers, exponents;
  std::vector<double> patches;
  std::vector<uint32_t> map_groups;
  uint8_t variant = 0;
  uint32_t exc = 0, clean_run = 0;
  const double limit = std::ldexp(1.0, (int)cx.opt.decimal_bits_limit);
  const uint32_t groups = (n + DEC_GROUP_ROWS - 1) / DEC_GROUP_ROWS;
  for (uint32_t g = 0; g < groups; g++) {
    bool dirty = false;
    const uint32_t end = std::min(n, (g + 1) * DEC_GROUP_ROWS);
    for (uint32_t i = g * DEC_GROUP_ROWS; i < end; i++) {
      const double x = src[i];
      bool ok = false;
      uint32_t e = 0;
      uint64_t digits = 0;
      if (!(x == 0.0 && std::signbit(x))) {
        for (e = 0; e <= DEC_MAX_EXPONENT; e++) {
          const double cd = std::round(x / kFractions[e]);
          if (!(cd >= 0.0) || cd >= 9223372036854775808.0) continue;  // not representable as u64 digits
          digits = (uint64_t)cd;
          if ((double)digits * kFractions[e] == x && cd < limit) {
            ok = true;
            break;
          }
        }
      }
      if (ok) {
        exponents.push_back((int32_t)e);
        numbers.push_back((int32_t)digits);
      } else {
        dirty = true;
        if (++exc > n / 2) return false;
        exponents.push_back((int32_t)DEC_EXCEPTION_CODE);
        patches.push_back(x);
      }
    }
    if (dirty) {
      clean_run = 0;
      map_groups.push_back(g);
    } else {
      clean_run++;
      variant |= 1;
      if (clean_run >= 4) variant |= 2;
    }
  }
  const size_t hdr = w.reserve(DEC_DATA);
  const size_t data = w.pos();
  wr32(w.at(hdr + DEC_CONVERTED), (uint32_t)numbers.size());
  *w.at(hdr + DEC_VARIANT) = variant;
  if (!numbers.empty()) {
    const uint8_t c = compress_int(w, cx, numbers.data(), nullptr, (uint32_t)numbers.size(),
                                   level - 1, spec.kid(0), false);
    *w.at(hdr + DEC_NUMBERS_SCHEME) = c;
  }
  wr32(w.at(hdr + DEC_EXPONENTS_OFFSET), (uint32_t)(w.pos() - data));
  {
    const uint8_t c = compress_int(w, cx, exponents.data(), nullptr, (uint32_t)exponents.size(),
                                   level - 1, spec.kid(1), false);
    *w.at(hdr + DEC_EXPONENTS_SCHEME) = c;
  }
  wr32(w.at(hdr + DEC_PATCHES_OFFSET), (uint32_t)(w.pos() - data));
  {
    const uint8_t c = compress_dbl(w, cx, patches.data(), nullptr, (uint32_t)patches.size(),
                                   level - 1, spec.kid(2), false);
    *w.at(hdr + DEC_PATCHES_SCHEME) = c;
  }
  wr32(w.at(hdr + DEC_MAP_OFFSET), (uint32_t)(w.pos() - data));
  roaring_serialize(map_groups.data(), map_groups.size(), w.buf);
  return true;
}

// ---- scheme selection ----------------------------------------------------------------------
template <typename T>
bool scheme_applicable(bool is_double, uint8_t code, const Stats<T>& st, int level,
                       const Options& o) {
  (void)o;
  if (is_double) {
    switch (code) {
      case D_DICT: return level > 1;  // DynamicDictionary.hpp:19-21
      case D_RLE: return st.avg_run_length >= 2;  // double/RLE.cpp:17-19
      case D_FREQUENCY: {  // Frequency.hpp:23-48 + double/Frequency.cpp:21-25
        if (st.distinct.size() <= 1) return false;
        if ((double)st.distinct.size() * 100.0 / (double)st.n > 50.0) return false;
        uint32_t best = 0;
        for (auto c : st.counts) best = std::max(best, c);
        return (double)st.null_count * 100.0 / st.n >= 90 ||
               ((double)best * 100.0 / st.n >= 90 && st.max >= 256);
      }
      case D_PSEUDODECIMAL:  // Pseudodecimal.cpp:476-483
        return (double)st.distinct.size() / (double)st.n >= 0.1;
      default: return true;
    }
  }
  switch (code) {
    case I_DICT: return level > 1;
    case I_RLE: return st.avg_run_length >= 2;  // integer/RLE.cpp:12-14
    default: return true;
  }
}

template <typename T>
uint8_t pick_scheme(const Ctx& cx, const T* src, const uint8_t* nm, uint32_t n, int level,
                    const Stats<T>& st) {
  constexpr bool is_double = sizeof(T) == 8;
  static const uint8_t kInt[] = {I_DICT, I_RLE, I_PFOR, I_BP};
  static const uint8_t kDbl[] = {D_DICT, D_RLE, D_FREQUENCY, D_PSEUDODECIMAL};
  const uint8_t* cand = is_double ? kDbl : kInt;
  std::vector<T> sv;
  std::vector<uint8_t> sn;
  const T* data = src;
  const uint8_t* dnm = nm;
  uint32_t dn = n;
  if (!cx.estimating) {
    take_sample(src, nm, n, cx.opt, sv, sn);
    data = sv.data();
    dnm = sn.data();
    dn = (uint32_t)sv.size();
  }
  const Ctx inner{cx.opt, true};
  uint8_t best = is_double ? (uint8_t)D_UNCOMPRESSED : (uint8_t)I_UNCOMPRESSED;
  double best_ratio = 1.0;
  for (int k = 0; k < 4; k++) {
    if (!scheme_applicable(is_double, cand[k], st, level, cx.opt)) continue;
    std::vector<uint8_t> tmp;
    Writer tw{tmp, 0};
    Spec forced;
    forced.scheme = cand[k];
    uint8_t got;
    if constexpr (is_double)
      got = compress_dbl(tw, inner, data, dnm, dn, level, forced, false);
    else
      got = compress_int(tw, inner, data, dnm, dn, level, forced, false);
    if (got != cand[k] || tmp.empty()) continue;
    const double ratio = (double)dn * sizeof(T) / (double)tmp.size();
    if (ratio > best_ratio) best_ratio = ratio, best = cand[k];
  }
  return best;
}

uint8_t compress_int(Writer& w, const Ctx& cx, const int32_t* src, const uint8_t* nm, uint32_t n,
                     int level, const Spec& spec, bool top) {
  const bool forced = spec.scheme != SCHEME_AUTO;
  const size_t start = w.pos();
  if (level <= 0 || n == 0) {  // SchemePicker.hpp:124-128
    w.put(src, (size_t)n * 4);
    return I_UNCOMPRESSED;
  }
  Stats<int32_t> st = make_stats(src, nm, n);
  uint8_t code;
  if (!forced && top && !cx.estimating && (st.null_count == n || st.distinct.size() == 1))
    code = I_ONE_VALUE;  // SchemePicker.hpp:118-123
  else if (!forced && cx.opt.enable_for && bits_needed_s32(st.min) >= 8 && level > 1)
    code = I_FOR;  // SchemePicker.hpp:131-136
  else
    code = forced ? spec.scheme : pick_scheme(cx, src, nm, n, level, st);

  auto emit = [&](uint8_t c) -> bool {
    switch (c) {
      case I_UNCOMPRESSED: w.put(src, (size_t)n * 4); return true;
      case I_ONE_VALUE: {
        const int32_t v = st.distinct[0];
        w.put(&v, 4);
        return true;
      }
      case I_DICT: emit_dict(w, cx, src, n, level, spec, st); return true;
      case I_RLE: emit_rle(w, cx, src, nm, n, level, spec); return true;
      case I_BP: emit_fastpfor(w, (const uint32_t*)src, n, false); return true;
      case I_PFOR: emit_fastpfor(w, (const uint32_t*)src, n, true); return true;
      case I_PFOR_DELTA: {
        std::vector<uint32_t> d(n);
        for (uint32_t i = 0; i < n; i++)
          d[i] = (uint32_t)src[i] - (i ? (uint32_t)src[i - 1] : 0u);
        // PBP_DELTA aligns to 4, not 16, and reports no slack (PBP.cpp:79-89).
        const size_t hdr = w.reserve(BP_DATA);
        const size_t pad = w.pad_to(4);
        w.reserve(pad);
        std::vector<uint32_t> words;
        pfor_encode(d.data(), n, words, ((w.pos() - w.origin) / 4) % 4);
        w.put(words.data(), words.size() * 4);
        wr32(w.at(hdr + BP_WORDS), (uint32_t)words.size());
        *w.at(hdr + BP_PADDING) = (uint8_t)pad;
        return true;
      }
      case I_FREQUENCY: emit_frequency(w, cx, src, nm, n, level, spec, st); return true;
      case I_FOR: {
        const size_t hdr = w.reserve(FOR_DATA);
        const int32_t bias = st.min;
        std::vector<int32_t> biased(n);
        for (uint32_t i = 0; i < n; i++)
          biased[i] = (nm == nullptr || nm[i]) ? (int32_t)((uint32_t)src[i] - (uint32_t)bias) : src[i];
        memcpy(w.at(hdr + FOR_BIAS), &bias, 4);
        const uint8_t c2 = compress_int(w, cx, biased.data(), nm, n, level - 1, spec.kid(0), false);
        *w.at(hdr + FOR_NEXT) = c2;
        return true;
      }
      case I_DICTIONARY_8:
        if (st.distinct.size() > 256) return false;
        emit_fixed_dict<int32_t, uint8_t>(w, src, n, st);
        return true;
      case I_DICTIONARY_16:
        if (st.distinct.size() > 65536) return false;
        emit_fixed_dict<int32_t, uint16_t>(w, src, n, st);
        return true;
      default: return false;
    }
  };
  const bool ok = emit(code);
  // Larger than raw (or not encodable): fall back to UNCOMPRESSED (SchemePicker.hpp:181-190).
  if (!ok || w.pos() - start > (size_t)n * 4) {
    w.buf.resize(start);
    w.put(src, (size_t)n * 4);
    return I_UNCOMPRESSED;
  }
  return code;
}

uint8_t compress_dbl(Writer& w, const Ctx& cx, const double* src, const uint8_t* nm, uint32_t n,
                     int level, const Spec& spec, bool top) {
  const bool forced = spec.scheme != SCHEME_AUTO;
  const size_t start = w.pos();
  if (level <= 0 || n == 0) {
    w.put(src, (size_t)n * 8);
    return D_UNCOMPRESSED;
  }
  Stats<double> st = make_stats(src, nm, n);
  uint8_t code;
  if (!forced && top && !cx.estimating && (st.null_count == n || st.distinct.size() == 1))
    code = D_ONE_VALUE;
  else
    code = forced ? spec.scheme : pick_scheme(cx, src, nm, n, level, st);

  auto emit = [&](uint8_t c) -> bool {
    switch (c) {
      case D_UNCOMPRESSED: w.put(src, (size_t)n * 8); return true;
      case D_ONE_VALUE: {
        const double v = st.distinct[0];
        w.put(&v, 8);
        return true;
      }
      case D_DICT: emit_dict(w, cx, src, n, level, spec, st); return true;
      case D_RLE: emit_rle(w, cx, src, nm, n, level, spec); return true;
      case D_FREQUENCY: emit_frequency(w, cx, src, nm, n, level, spec, st); return true;
      case D_PSEUDODECIMAL: return emit_pseudodecimal(w, cx, src, n, level, spec);
      case D_DOUBLE_BP: emit_fastpfor(w, (const uint32_t*)src, (size_t)n * 2, false); return true;
      case D_DICTIONARY_8:
        if (st.distinct.size() > 256) return false;
        emit_fixed_dict<double, uint8_t>(w, src, n, st);
        return true;
      case D_DICTIONARY_16:
        if (st.distinct.size() > 65536) return false;
        emit_fixed_dict<double, uint16_t>(w, src, n, st);
        return true;
      default: return false;
    }
  };
  const bool ok = emit(code);
  if (!ok || w.pos() - start > (size_t)n * 8) {
    w.buf.resize(start);
    w.put(src, (size_t)n * 8);
    return D_UNCOMPRESSED;
  }
  return code;
}

// nullmap: ALLONES / ALLZEROS carry no bytes; otherwise the rarer bit value's positions go into
// a Roaring bitmap (RoaringBitmap.cpp:117-159).
uint8_t emit_nullmap(Writer& w, const uint8_t* nm, uint32_t n) {
  if (nm == nullptr) return NM_ALLONES;
  uint32_t ones = 0;
  for (uint32_t i = 0; i < n; i++) ones += nm[i];
  if (ones == 0) return NM_ALLZEROS;
  if (ones == n) return NM_ALLONES;
  const bool regular = ones < n / 2;
  const uint8_t want = regular ? 1 : 0;
  std::vector<uint32_t> rows;
  for (uint32_t i = 0; i < n; i++)
    if (nm[i] == want) rows.push_back(i);
  roaring_serialize(rows.data(), rows.size(), w.buf);
  return regular ? NM_REGULAR : NM_FLIPPED;
}
}  // namespace

size_t compress_chunk(uint8_t col_type, const void* values, const uint8_t* nullmap, uint32_t n,
                      const Spec& spec, const Options& opt, std::vector<uint8_t>& out) {
  while (out.size() % PART_ALIGN) out.push_back(0);
  const size_t origin = out.size();
  Writer w{out, origin};
  w.reserve(CHUNK_DATA);
  const Ctx cx{opt, false};
  uint8_t code;
  if (col_type == COL_INTEGER)
    code = compress_int(w, cx, (const int32_t*)values, nullptr == nullmap ? nullptr : nullmap, n,
                        opt.max_cascade_depth, spec, true);
  else if (col_type == COL_DOUBLE)
    code = compress_dbl(w, cx, (const double*)values, nullmap, n, opt.max_cascade_depth, spec, true);
  else
    throw std::invalid_argument("only integer and double columns are supported");
  const uint32_t nm_off = (uint32_t)(w.pos() - origin - CHUNK_DATA);
  const uint8_t nm_type = emit_nullmap(w, nullmap, n);
  uint8_t* h = w.at(origin);
  h[CHUNK_SCHEME] = code;
  h[CHUNK_NM_TYPE] = nm_type;
  h[CHUNK_COL_TYPE] = col_type;
  wr32(h + CHUNK_NM_OFFSET, nm_off);
  wr32(h + CHUNK_TUPLES, n);
  return origin;
}

size_t compress_string_dict_chunk(const uint8_t* bytes, const uint32_t* offsets, uint32_t n,
                                  const uint8_t* nullmap, const StringDictOptions& sopt,
                                  const Options& opt, std::vector<uint8_t>& out) {
  if (n == 0) throw std::invalid_argument("a string chunk needs at least one row");
  using View = std::pair<const uint8_t*, uint32_t>;
  auto view = [&](uint32_t i) { return View{bytes + offsets[i], offsets[i + 1] - offsets[i]}; };
  auto less = [](const View& a, const View& b) {  // std::string_view order
    const int c = memcmp(a.first, b.first, std::min(a.second, b.second));
    return c < 0 || (c == 0 && a.second < b.second);
  };
  for (uint32_t i = 0; i < n; i++)
    if (offsets[i + 1] < offsets[i]) throw std::invalid_argument("string offsets must not decrease");
  std::vector<View> distinct(n);
  for (uint32_t i = 0; i < n; i++) distinct[i] = view(i);
  std::sort(distinct.begin(), distinct.end(), less);
  distinct.erase(std::unique(distinct.begin(), distinct.end(),
                             [&](const View& a, const View& b) { return !less(a, b) && !less(b, a); }),
                 distinct.end());
  const uint32_t cardinality = (uint32_t)distinct.size();

  while (out.size() % PART_ALIGN) out.push_back(0);
  const size_t origin = out.size();
  Writer w{out, origin};
  w.reserve(CHUNK_DATA);
  const size_t hdr = w.reserve(SDICT_DATA);
  const size_t data = w.pos();
  Options inner_opt = opt;
  inner_opt.max_cascade_depth = sopt.codes_cascade_depth;
  const Ctx cx{inner_opt, false};
  uint8_t lengths_code = 0;
  uint32_t fsst_offsets_offset = 0, lengths_offset = 0;
  if (sopt.fsst_layout) {
    w.reserve(SDICT_FSST_HEADER);
    std::vector<uint32_t> pool_offsets(cardinality + 1);
    std::vector<int32_t> lengths(cardinality);
    uint32_t at = 0;
    for (uint32_t c = 0; c < cardinality; c++) {
      pool_offsets[c] = at;
      w.put(distinct[c].first, distinct[c].second);
      lengths[c] = (int32_t)distinct[c].second;
      at += distinct[c].second;
    }
    pool_offsets[cardinality] = at;
    fsst_offsets_offset = (uint32_t)(w.pos() - data);
    w.put(pool_offsets.data(), pool_offsets.size() * 4);
    lengths_offset = (uint32_t)(w.pos() - data);
    lengths_code = compress_int(w, cx, lengths.data(), nullptr, cardinality, sopt.codes_cascade_depth,
                                sopt.lengths_spec, false);
  } else {
    // StringArrayViewer: (cardinality + 1) u32 slots counted from the first slot, then the bytes
    const size_t slots = w.reserve((size_t)(cardinality + 1) * 4);
    for (uint32_t c = 0; c < cardinality; c++) {
      wr32(w.at(slots + 4 * (size_t)c), (uint32_t)(w.pos() - data));
      w.put(distinct[c].first, distinct[c].second);
    }
    wr32(w.at(slots + 4 * (size_t)cardinality), (uint32_t)(w.pos() - data));
  }
  std::vector<int32_t> codes(n);
  uint32_t runs = 0;
  int32_t current = -1;
  for (uint32_t i = 0; i < n; i++) {
    codes[i] = (int32_t)(std::lower_bound(distinct.begin(), distinct.end(), view(i), less) - distinct.begin());
    if (codes[i] != current) {
      runs++;
      current = codes[i];
    }
  }
  Spec codes_spec = sopt.codes_spec;
  bool use_rle = (double)n / (double)runs > 3.0;
  if (use_rle && codes_spec.scheme != I_RLE) {
    codes_spec = Spec{};
    codes_spec.scheme = I_RLE;
  }
  const uint32_t codes_offset = (uint32_t)(w.pos() - data);
  const uint8_t codes_code = compress_int(w, cx, codes.data(), nullptr, n, sopt.codes_cascade_depth,
                                          codes_spec, false);
  // A tiny block whose RLE form exceeds the raw codes falls back to UNCOMPRESSED; the reference
  // would then cast the wrong scheme object (:242), so the flag is only kept for a real RLE node.
  if (codes_code != I_RLE) use_rle = false;
  uint8_t* s = w.at(hdr);
  wr32(s + SDICT_TOTAL_SIZE, offsets[n] - offsets[0] + (n + 1) * 4);  // slots + bytes
  s[SDICT_USE_FSST] = sopt.fsst_layout ? 1 : 0;
  wr32(s + SDICT_FSST_OFFSETS_OFFSET, fsst_offsets_offset);
  wr32(s + SDICT_LENGTHS_OFFSET, lengths_offset);
  wr32(s + SDICT_NUM_CODES, cardinality);
  wr32(s + SDICT_CODES_OFFSET, codes_offset);
  s[SDICT_USE_RLE] = use_rle ? 1 : 0;
  s[SDICT_CODES_SCHEME] = codes_code;
  s[SDICT_LENGTHS_SCHEME] = lengths_code;
  const uint32_t nm_off = (uint32_t)(w.pos() - origin - CHUNK_DATA);
  const uint8_t nm_type = emit_nullmap(w, nullmap, n);
  uint8_t* h = w.at(origin);
  h[CHUNK_SCHEME] = S_DICT;
  h[CHUNK_NM_TYPE] = nm_type;
  h[CHUNK_COL_TYPE] = COL_STRING;
  wr32(h + CHUNK_NM_OFFSET, nm_off);
  wr32(h + CHUNK_TUPLES, n);
  return origin;
}

void compress_datablock(const std::vector<ColumnInput>& cols, uint32_t n, const Options& opt,
                        std::vector<uint8_t>& out) {
  while (out.size() % PART_ALIGN) out.push_back(0);
  const size_t origin = out.size();
  Writer w{out, origin};
  w.reserve(DB_ATTRS + DB_ATTR_BYTES * cols.size());
  const Ctx cx{opt, false};
  for (size_t c = 0; c < cols.size(); c++) {
    const size_t col_start = w.pos();
    uint8_t code;
    if (cols[c].col_type == COL_INTEGER) {
      code = compress_int(w, cx, (const int32_t*)cols[c].values, cols[c].nullmap, n,
                          opt.max_cascade_depth, cols[c].spec, true);
      w.reserve(4);  // Datablock.cpp:243 adds sizeof(bias) to every integer column
    } else if (cols[c].col_type == COL_DOUBLE) {
      code = compress_dbl(w, cx, (const double*)cols[c].values, cols[c].nullmap, n,
                          opt.max_cascade_depth, cols[c].spec, true);
    } else {
      throw std::invalid_argument("only integer and double columns are supported");
    }
    const size_t nm_start = w.pos();
    const uint8_t nm_type = emit_nullmap(w, cols[c].nullmap, n);
    uint8_t* m = w.at(origin + DB_ATTRS + DB_ATTR_BYTES * c);
    m[CM_SCHEME] = code;
    m[CM_COL_TYPE] = cols[c].col_type;
    m[CM_NM_TYPE] = nm_type;
    wr32(m + CM_OFFSET, (uint32_t)(col_start - origin));
    wr32(m + CM_BIAS, 0);
    wr32(m + CM_NM_OFFSET, (uint32_t)(nm_start - origin));
  }
  uint8_t* h = w.at(origin);
  wr32(h + DB_COUNT, n);
  wr32(h + DB_SIZE, (uint32_t)(w.pos() - origin));
  wr32(h + DB_COLUMNS, (uint32_t)cols.size());
}

void build_part_image(const std::vector<std::vector<uint8_t>>& chunks, std::vector<uint8_t>& out) {
  out.clear();
  out.resize(4 + 4 * chunks.size(), 0);
  wr32(out.data(), (uint32_t)chunks.size());
  for (size_t i = 0; i < chunks.size(); i++) {
    while (out.size() % PART_ALIGN) out.push_back(0);
    wr32(out.data() + PART_OFFSETS + 4 * i, (uint32_t)out.size());
    out.insert(out.end(), chunks[i].begin(), chunks[i].end());
  }
}

}  // namespace btrhost
