// Host-side (CPU) compressor producing BtrBlocks column chunks and relation blocks.
// Compression-side scheme selection stays on the CPU by design; the GPU only decodes.
//
// A `Spec` names the cascade to emit: either a forced tree such as rle(dict(bp),bp) or `auto`,
// which runs a deterministic sample-based picker in the spirit of
// btrblocks/compression/SchemePicker.hpp:70-221 (the reference samples at random positions,
// so its choices are not reproducible; this picker samples at fixed, evenly spaced positions).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace btrhost {

struct Spec {
  uint8_t scheme = 255;  // wire code, or 255 = pick automatically
  std::vector<Spec> kids;
  const Spec& kid(size_t i) const;
};

// Grammar: name | name(child{,child}); names (case-insensitive): auto, uncompressed, onevalue,
// dict, rle, pfor, bp, frequency, for, pfor_delta, dict8, dict16, pseudodecimal, doublebp.
// Child order: rle(values,counts) dict(codes) for(next) frequency(exceptions)
// pseudodecimal(digits,exponents,patches).
bool parse_spec(const char* text, bool is_double, Spec& out, std::string& err);

struct Options {
  uint8_t max_cascade_depth = 3;  // BtrBlocksConfig default, btrblocks/btrblocks.hpp:31,37
  bool enable_for = false;        // ENABLE_FOR_SCHEME build flag, SchemePicker.hpp:13-23
  uint32_t sample_count = 10;     // btrblocks/btrblocks.hpp:25-26
  uint32_t sample_size = 64;
  uint32_t decimal_bits_limit = 31;  // SchemeConfig.hpp:41
};

// Appends one column chunk (12-byte ColumnChunkMeta envelope + scheme payload + nullmap) to `out`.
// The chunk start is padded to a multiple of 16 bytes of out.size() first, as column part files
// do (btrblocks/storage/Chunk.cpp:261-272), so FastPFor payload alignment is reproducible.
// Returns the offset of the chunk inside `out`. `nullmap` may be null (all rows valid).
size_t compress_chunk(uint8_t col_type, const void* values, const uint8_t* nullmap, uint32_t n,
                      const Spec& spec, const Options& opt, std::vector<uint8_t>& out);

// String column chunk in the DICT scheme (btrblocks/scheme/string/DynamicDictionary.cpp:42-208):
// sorted distinct strings, then the codes as an integer stream. `offsets` holds n + 1 byte
// offsets into `bytes`. The codes are forced to RLE (use_rle_optimized_path) when the average run
// is longer than 3 rows (:183-189); otherwise `codes_spec` applies. With `fsst_layout` the pool is
// laid out as the FSST variant is (symbol-table area, string bytes, u32 offsets, lengths stream,
// :89-152) but the strings are stored verbatim and the symbol-table area is zeroed: such a chunk
// exercises the lengths sub-stream and is not readable by an FSST decoder.
struct StringDictOptions {
  Spec codes_spec;
  Spec lengths_spec;
  bool fsst_layout = false;
  uint8_t codes_cascade_depth = 2;  // fsst_codes_max_cascade_depth, SchemeConfig.hpp:57
};
size_t compress_string_dict_chunk(const uint8_t* bytes, const uint32_t* offsets, uint32_t n,
                                  const uint8_t* nullmap, const StringDictOptions& sopt,
                                  const Options& opt, std::vector<uint8_t>& out);

// Relation-level block (DatablockMeta + ColumnMeta[] + payloads), the format of
// Datablock::compress(const Chunk&, BytesArray&) (btrblocks/compression/Datablock.cpp:192-338).
struct ColumnInput {
  uint8_t col_type;
  const void* values;
  const uint8_t* nullmap;  // may be null
  Spec spec;
};
void compress_datablock(const std::vector<ColumnInput>& cols, uint32_t n, const Options& opt,
                        std::vector<uint8_t>& out);

// Column part image: {u32 num_chunks, u32 offsets[]} + 16-byte aligned chunks
// (btrblocks/storage/Chunk.cpp:250-295).
void build_part_image(const std::vector<std::vector<uint8_t>>& chunks, std::vector<uint8_t>& out);

}  // namespace btrhost
