// btrblocks_amd.hpp - C++ host interface over the C ABI (btrblocks_amd.h).
//
// Mirrors the part of the reference's C++ API that its own example and tools use for the
// compress / decompress round trip of integer and double columns
//   Relation, Column, Range, SplitStrategy      btrblocks/storage/Relation.hpp:7-35, Column.hpp
//   Chunk                                       btrblocks/storage/Chunk.hpp:20-60
//   Datablock::compress / decompress            btrblocks/compression/Datablock.hpp:57-72
//   BtrReader                                   btrblocks/compression/BtrReader.hpp:12-45
// with the same names and argument meaning, so code written against the reference (e.g.
// tools/examples/compression.cpp:56-117) ports by switching the namespace. Compression runs on
// the CPU, decompression on the GPU; a failing call throws Generic_Exception like the reference
// (common/Exceptions.hpp:15-26). Header only; link with -lbtrblocks_amd.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "btrblocks_amd.h"

namespace btrblocks_amd {

using u8 = uint8_t;
using u32 = uint32_t;
using u64 = uint64_t;
using BITMAP = u8;  // byte per row, 1 = valid (common/Units.hpp)
using BytesArray = std::unique_ptr<u8[]>;
using Range = std::tuple<u64, u64>;  // first row, row count

enum class ColumnType : u8 { INTEGER = 0, DOUBLE = 1, STRING = 2 };
enum class SplitStrategy : u8 { SEQUENTIAL, RANDOM };

struct Generic_Exception : std::runtime_error {
  btr_status status;
  Generic_Exception(btr_status st, const std::string& what) : std::runtime_error(what), status(st) {}
};
inline void check(btr_status st) {
  if (st != BTR_OK) throw Generic_Exception(st, btr_last_error());
}

// A column of a relation: values plus an optional byte-per-row validity map.
struct Column {
  std::string name;
  ColumnType type;
  std::vector<int32_t> ints;
  std::vector<double> doubles;
  std::vector<BITMAP> bitmap;  // empty = all rows valid

  Column(std::string n, std::vector<int32_t> v, std::vector<BITMAP> b = {})
      : name(std::move(n)), type(ColumnType::INTEGER), ints(std::move(v)), bitmap(std::move(b)) {}
  Column(std::string n, std::vector<double> v, std::vector<BITMAP> b = {})
      : name(std::move(n)), type(ColumnType::DOUBLE), doubles(std::move(v)), bitmap(std::move(b)) {}
  u64 size() const { return type == ColumnType::INTEGER ? ints.size() : doubles.size(); }
  const void* data() const { return type == ColumnType::INTEGER ? (const void*)ints.data() : (const void*)doubles.data(); }
  size_t value_bytes() const { return type == ColumnType::INTEGER ? 4 : 8; }
};

// A horizontal slice of a relation (or a decompressed block): one array and one bitmap per column.
struct Chunk {
  u64 tuple_count = 0;
  std::vector<ColumnType> column_types;
  std::vector<std::unique_ptr<u8[]>> columns;
  std::vector<std::unique_ptr<BITMAP[]>> bitmaps;
  std::vector<size_t> sizes;  // bytes per column

  template <typename T>
  const T* array(u32 col) const { return reinterpret_cast<const T*>(columns[col].get()); }
  const BITMAP* nullmap(u32 col) const { return bitmaps[col].get(); }
  bool operator==(const Chunk& o) const {
    if (tuple_count != o.tuple_count || columns.size() != o.columns.size()) return false;
    for (size_t c = 0; c < columns.size(); c++) {
      if (sizes[c] != o.sizes[c] || std::memcmp(bitmaps[c].get(), o.bitmaps[c].get(), tuple_count)) return false;
      // rows behind a zero in the bitmap are not part of the contract (Chunk.cpp:60-105)
      const size_t w = sizes[c] / (tuple_count ? tuple_count : 1);
      for (u64 i = 0; i < tuple_count; i++)
        if (bitmaps[c][i] && std::memcmp(columns[c].get() + i * w, o.columns[c].get() + i * w, w)) return false;
    }
    return true;
  }
};

class Relation {
 public:
  std::string name;
  u64 tuple_count = 0;
  std::vector<Column> columns;

  void addColumn(Column&& column) {
    if (!columns.empty() && column.size() != tuple_count)
      throw Generic_Exception(BTR_ERR_INVALID_ARGUMENT, "columns of a relation must have the same length");
    tuple_count = column.size();
    columns.push_back(std::move(column));
  }
  // Row ranges of at most max_chunk_count rows each (storage/Relation.cpp:42-58).
  std::vector<Range> getRanges(SplitStrategy, u32 max_chunk_count) const {
    std::vector<Range> r;
    for (u64 at = 0; at < tuple_count; at += max_chunk_count)
      r.emplace_back(at, std::min<u64>(max_chunk_count, tuple_count - at));
    return r;
  }
  Chunk getChunk(const std::vector<Range>& ranges, size_t chunk_i) const {
    const u64 first = std::get<0>(ranges.at(chunk_i)), n = std::get<1>(ranges.at(chunk_i));
    Chunk c;
    c.tuple_count = n;
    for (const Column& col : columns) {
      const size_t w = col.value_bytes();
      std::unique_ptr<u8[]> d(new u8[n * w + 1]);
      std::memcpy(d.get(), (const u8*)col.data() + first * w, n * w);
      std::unique_ptr<BITMAP[]> b(new BITMAP[n + 1]);
      if (col.bitmap.empty())
        std::memset(b.get(), 1, n);
      else
        std::memcpy(b.get(), col.bitmap.data() + first, n);
      c.column_types.push_back(col.type);
      c.columns.push_back(std::move(d));
      c.bitmaps.push_back(std::move(b));
      c.sizes.push_back(n * w);
    }
    return c;
  }
};

struct OutputBlockStats {
  size_t total_data_size = 0;  // compressed block bytes
  size_t total_uncompressed = 0;
  double compression_ratio = 0;
  std::vector<std::string> used_compression_schemes;  // scheme tree per column
};

// Relation-level block compressor / decompressor. `scheme_spec` forces a cascade for every column
// ("auto" = sample-based selection, the reference's default).
class Datablock {
 public:
  explicit Datablock(const Relation& relation, int device = 0, std::string scheme_spec = "auto")
      : spec_(std::move(scheme_spec)) {
    for (const Column& c : relation.columns) types_.push_back(c.type);
    check(btr_gpu_ctx_create(device, &ctx_));
  }
  ~Datablock() { btr_gpu_ctx_destroy(ctx_); }
  Datablock(const Datablock&) = delete;
  Datablock& operator=(const Datablock&) = delete;

  OutputBlockStats compress(const Chunk& input, BytesArray& output) {
    std::vector<btr_column_input> cols(input.columns.size());
    size_t bound = 64 + 16 * cols.size(), raw = 0;
    for (size_t c = 0; c < cols.size(); c++) {
      cols[c] = {(u8)input.column_types[c], input.columns[c].get(), input.bitmaps[c].get(), spec_.c_str()};
      bound += btr_compress_bound((u8)input.column_types[c], (u32)input.tuple_count) + 16;
      raw += input.sizes[c];
    }
    output.reset(new u8[bound]);
    size_t bytes = 0;
    check(btr_compress_datablock(cols.data(), (u32)cols.size(), (u32)input.tuple_count, nullptr, output.get(), bound, &bytes));
    size_ = bytes;
    OutputBlockStats st;
    st.total_data_size = bytes;
    st.total_uncompressed = raw;
    st.compression_ratio = bytes ? (double)raw / (double)bytes : 0;
    return st;
  }
  size_t lastBlockSize() const { return size_; }

  // Decodes every column of a block on the GPU (Datablock.cpp:340-417) and returns host arrays.
  Chunk decompress(const BytesArray& input_block) { return decompress(input_block.get(), size_); }
  Chunk decompress(const u8* block, size_t bytes) {
    u32 n = 0, cols = 0;
    check(btr_datablock_inspect(block, bytes, &n, &cols));
    Chunk out;
    out.tuple_count = n;
    void* d_block = nullptr;
    std::vector<void*> d_cols(cols, nullptr);
    std::vector<u8*> d_maps(cols, nullptr);
    btr_gpu_plan* plan = nullptr;
    auto release = [&]() {
      if (plan) btr_gpu_plan_destroy(plan);
      for (void* p : d_cols) btr_gpu_free(ctx_, p);
      for (u8* p : d_maps) btr_gpu_free(ctx_, p);
      btr_gpu_free(ctx_, d_block);
    };
    try {
      check(btr_gpu_malloc(ctx_, bytes + 16, &d_block));
      check(btr_gpu_upload(ctx_, d_block, block, bytes));
      std::vector<btr_chunk_info> info(cols);
      for (u32 c = 0; c < cols; c++) {
        check(btr_datablock_column(block, bytes, c, &info[c]));
        check(btr_gpu_malloc(ctx_, info[c].decoded_bytes + 16, &d_cols[c]));
        void* m = nullptr;
        check(btr_gpu_malloc(ctx_, (size_t)n + 16, &m));
        d_maps[c] = (u8*)m;
      }
      check(btr_gpu_plan_create_datablock(ctx_, block, d_block, bytes, d_cols.data(), d_maps.data(), cols, &plan));
      check(btr_gpu_decode(ctx_, plan, nullptr));
      check(btr_gpu_plan_check(ctx_, plan, nullptr, nullptr, nullptr));
      for (u32 c = 0; c < cols; c++) {
        std::unique_ptr<u8[]> d(new u8[info[c].decoded_bytes + 1]);
        std::unique_ptr<BITMAP[]> b(new BITMAP[n + 1]);
        check(btr_gpu_download(ctx_, d.get(), d_cols[c], info[c].decoded_bytes));
        check(btr_gpu_download(ctx_, b.get(), d_maps[c], n));
        out.column_types.push_back((ColumnType)info[c].column_type);
        out.columns.push_back(std::move(d));
        out.bitmaps.push_back(std::move(b));
        out.sizes.push_back(info[c].decoded_bytes);
      }
    } catch (...) {
      release();
      throw;
    }
    release();
    return out;
  }

 private:
  std::vector<ColumnType> types_;
  std::string spec_;
  btr_gpu_ctx* ctx_ = nullptr;
  size_t size_ = 0;
};

// Reader over one column part image {u32 num_chunks; u32 offsets[]} + chunks in host memory
// (compression/BtrReader.hpp:12-45). The image is uploaded once. The first readColumn / getBitmap
// decodes every chunk of the part in one launch into a device-resident column (one workgroup per
// chunk); after that readColumn(i) only copies chunk i back to the host.
class BtrReader {
 public:
  BtrReader(const void* data, size_t bytes, int device = 0) : data_((const u8*)data), bytes_(bytes) {
    check(btr_gpu_ctx_create(device, &ctx_));
    u32 n = 0;
    check(btr_part_chunk_count(data_, bytes_, &n));
    infos_.resize(n);
    offsets_.resize(n);
    lengths_.resize(n);
    check(btr_part_inspect(data_, bytes_, 0, n, infos_.data(), offsets_.data(), lengths_.data()));
    out_off_.assign(n + 1, 0);
    row_off_.assign(n + 1, 0);
    for (u32 i = 0; i < n; i++) {
      out_off_[i + 1] = out_off_[i] + (infos_[i].decoded_bytes + 15) / 16 * 16;
      row_off_[i + 1] = row_off_[i] + infos_[i].tuple_count;
    }
    bitmaps_.resize(n);
    check(btr_gpu_malloc(ctx_, bytes_ + 16, &d_image_));
    check(btr_gpu_upload(ctx_, d_image_, data_, bytes_));
  }
  ~BtrReader() {
    if (plan_) btr_gpu_plan_destroy(plan_);
    btr_gpu_free(ctx_, d_out_);
    btr_gpu_free(ctx_, d_maps_);
    btr_gpu_free(ctx_, d_image_);
    btr_gpu_ctx_destroy(ctx_);
  }
  BtrReader(const BtrReader&) = delete;
  BtrReader& operator=(const BtrReader&) = delete;

  // BtrReader.hpp:20-39
  u32 getChunkCount() const { return (u32)infos_.size(); }
  u32 getTupleCount(u32 index) const { return infos_.at(index).tuple_count; }
  ColumnType getColumnType() const { return infos_.empty() ? ColumnType::INTEGER : (ColumnType)infos_[0].column_type; }
  size_t getDecompressedSize(u32 index) const { return infos_.at(index).decoded_bytes; }
  // BtrReader.cpp:186-207: for numbers the value bytes are the whole decoded chunk.
  size_t getDecompressedDataSize(u32 index) const { return infos_.at(index).decoded_bytes; }
  std::string getSchemeDescription(u32 index) const {
    char buf[512];
    check(btr_chunk_describe(data_ + offsets_.at(index), lengths_[index], buf, sizeof buf));
    return buf;
  }
  // Byte-per-row validity of chunk `index` (1 = valid), what BitmapWrapper::writeBITMAP produces
  // (BtrReader.cpp:124-141, extern/RoaringBitmap.cpp:32-53). Owned by the reader until
  // releaseBitmap(index) (BtrReader.cpp:143-157).
  const BITMAP* getBitmap(u32 index) {
    std::vector<BITMAP>& b = bitmaps_.at(index);
    if (b.empty() && infos_[index].tuple_count) {
      decodeAll(true);
      b.resize(infos_[index].tuple_count);
      check(btr_gpu_download(ctx_, b.data(), (const u8*)d_maps_ + row_off_[index], b.size()));
    }
    return b.data();
  }
  void releaseBitmap(u32 index) { std::vector<BITMAP>().swap(bitmaps_.at(index)); }

  // Decodes every chunk of the part in one launch (no-op when already done).
  void decodeAll(bool with_bitmaps = false) {
    if (plan_ && (!with_bitmaps || d_maps_)) return;
    if (plan_) {
      btr_gpu_plan_destroy(plan_);
      plan_ = nullptr;
    }
    if (!d_out_) check(btr_gpu_malloc(ctx_, out_off_.back() + 16, &d_out_));
    if (with_bitmaps && !d_maps_) check(btr_gpu_malloc(ctx_, row_off_.back() + 16, &d_maps_));
    check(btr_gpu_plan_create_part(ctx_, data_, d_image_, bytes_, 0, getChunkCount(), d_out_, out_off_.back() + 16,
                                   (u8*)d_maps_, d_maps_ ? row_off_.back() + 16 : 0, &plan_));
    check(btr_gpu_decode(ctx_, plan_, nullptr));
    check(btr_gpu_plan_check(ctx_, plan_, nullptr, nullptr, nullptr));
  }
  // Device address of decoded chunk `index` (after decodeAll).
  const void* deviceColumn(u32 index) const { return d_out_ ? (const u8*)d_out_ + out_off_.at(index) : nullptr; }

  // Chunk `index` into output_chunk_v (BtrReader.cpp:31-74); returns requires_copy (always
  // false: strings only). The vector is grown, never shrunk (common/Units.hpp:125-129).
  bool readColumn(std::vector<u8>& output_chunk_v, u32 index) {
    const size_t need = infos_.at(index).decoded_bytes;
    decodeAll(false);
    if (output_chunk_v.size() < need) output_chunk_v.resize(need);
    check(btr_gpu_download(ctx_, output_chunk_v.data(), (const u8*)d_out_ + out_off_[index], need));
    return false;
  }

 private:
  const u8* data_;
  size_t bytes_;
  btr_gpu_ctx* ctx_ = nullptr;
  void* d_image_ = nullptr;
  void* d_out_ = nullptr;
  void* d_maps_ = nullptr;
  btr_gpu_plan* plan_ = nullptr;
  std::vector<btr_chunk_info> infos_;
  std::vector<uint64_t> offsets_, lengths_, out_off_, row_off_;
  std::vector<std::vector<BITMAP>> bitmaps_;
};

}  // namespace btrblocks_amd
