// Wire format of BtrBlocks column chunks: byte offsets, scheme codes and small
// unaligned readers shared by the host plan builder, the host compressor and the
// HIP decode kernels. Every header is addressed by explicit byte offsets because
// scheme payloads start at odd offsets (5, 9, 10, 13, 20) inside a chunk.
//
// Reference layouts:
//   ColumnChunkMeta            btrblocks/compression/Datablock.hpp:12-21
//   ColumnPartMetadata         btrblocks/storage/Chunk.hpp:95-98
//   DatablockMeta / ColumnMeta btrblocks/compression/Datablock.hpp:37-55
//   RLEStructure               btrblocks/scheme/templated/RLE.hpp:8-14
//   DynamicDictionaryStructure btrblocks/scheme/templated/DynamicDictionary.hpp:7-11
//   XPBPStructure              btrblocks/scheme/integer/PBP.hpp:7-12
//   FORStructure               btrblocks/scheme/integer/FOR.hpp:8-12
//   FrequencyStructure<T>      btrblocks/scheme/templated/Frequency.hpp:12-18
//   DecimalStructure           btrblocks/scheme/double/Pseudodecimal.hpp:13-32
//   FixedDictionaryStructure   btrblocks/scheme/templated/FixedDictionary.hpp:7-11
//   string DICT structure      btrblocks/scheme/string/DynamicDictionary.hpp:7-18
//   scheme codes               btrblocks/scheme/SchemeType.hpp:12-28, 35-47
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define BTR_HD __host__ __device__ inline
#else
#define BTR_HD inline
#endif

namespace btrfmt {

// ---- column types (btrblocks/common/Units.hpp:47-57) and nullmap types (:45)
enum : uint8_t { COL_INTEGER = 0, COL_DOUBLE = 1, COL_STRING = 2 };
enum : uint8_t { NM_ALLONES = 0, NM_ALLZEROS = 1, NM_REGULAR = 2, NM_FLIPPED = 3 };

// ---- integer scheme wire codes
enum : uint8_t {
  I_UNCOMPRESSED = 0,
  I_ONE_VALUE = 1,
  I_DICT = 2,
  I_RLE = 3,
  I_PFOR = 4,
  I_BP = 5,
  I_FREQUENCY = 25,
  I_FOR = 26,
  I_PFOR_DELTA = 27,
  I_TRUNCATION_8 = 28,
  I_TRUNCATION_16 = 29,
  I_DICTIONARY_8 = 30,
  I_DICTIONARY_16 = 31,
};
// ---- double scheme wire codes
enum : uint8_t {
  D_UNCOMPRESSED = 0,
  D_ONE_VALUE = 1,
  D_DICT = 2,
  D_RLE = 3,
  D_FREQUENCY = 4,
  D_PSEUDODECIMAL = 5,
  D_DOUBLE_BP = 28,
  D_DICTIONARY_8 = 29,
  D_DICTIONARY_16 = 31,
};
// ---- string scheme wire codes (SchemeType.hpp:49-56); only DICT's integer sub-streams are decoded
enum : uint8_t { S_UNCOMPRESSED = 0, S_ONE_VALUE = 1, S_DICT = 2, S_FSST = 3 };
constexpr uint8_t SCHEME_AUTO = 255;  // autoScheme(), SchemeType.hpp:72-74

// ---- chunk envelope: {u8 scheme, u8 nullmap_type, u8 column_type, pad, u32 nullmap_offset, u32 tuple_count}
constexpr uint32_t CHUNK_SCHEME = 0, CHUNK_NM_TYPE = 1, CHUNK_COL_TYPE = 2, CHUNK_NM_OFFSET = 4,
                   CHUNK_TUPLES = 8, CHUNK_DATA = 12;
// ---- column part image: {u32 num_chunks, u32 offsets[]} then 16-byte aligned chunks
constexpr uint32_t PART_NUM_CHUNKS = 0, PART_OFFSETS = 4, PART_ALIGN = 16;
// ---- relation-level block: {u32 count, u32 size, u32 column_count, u32 pad, ColumnMeta[]}
constexpr uint32_t DB_COUNT = 0, DB_SIZE = 4, DB_COLUMNS = 8, DB_ATTRS = 16, DB_ATTR_BYTES = 16;
// ColumnMeta (packed): {u8 scheme, u8 column_type, u8 bitmap_type, u8 pad, u32 offset, s32 bias, u32 nullmap_offset}
constexpr uint32_t CM_SCHEME = 0, CM_COL_TYPE = 1, CM_NM_TYPE = 2, CM_OFFSET = 4, CM_BIAS = 8,
                   CM_NM_OFFSET = 12;

// ---- scheme headers
constexpr uint32_t RLE_RUNS = 0, RLE_COUNTS_OFFSET = 4, RLE_VALUES_SCHEME = 8, RLE_COUNTS_SCHEME = 9,
                   RLE_DATA = 10;
constexpr uint32_t DICT_CODES_SCHEME = 0, DICT_CODES_OFFSET = 1, DICT_DATA = 5;
constexpr uint32_t BP_WORDS = 0, BP_PADDING = 4, BP_DATA = 5;
constexpr uint32_t FOR_BIAS = 0, FOR_NEXT = 4, FOR_DATA = 5;
constexpr uint32_t FREQ_I_TOP = 0, FREQ_I_EXC_OFFSET = 4, FREQ_I_NEXT = 8, FREQ_I_DATA = 9;
constexpr uint32_t FREQ_D_TOP = 0, FREQ_D_EXC_OFFSET = 8, FREQ_D_NEXT = 12, FREQ_D_DATA = 13;
constexpr uint32_t DEC_CONVERTED = 0, DEC_NUMBERS_SCHEME = 4, DEC_EXPONENTS_SCHEME = 5,
                   DEC_PATCHES_SCHEME = 6, DEC_VARIANT = 7, DEC_EXPONENTS_OFFSET = 8,
                   DEC_PATCHES_OFFSET = 12, DEC_MAP_OFFSET = 16, DEC_DATA = 20;
constexpr uint32_t FDICT_CODES_OFFSET = 0, FDICT_I_SLOTS = 4, FDICT_D_SLOTS = 8;
// string DICT (not packed, so use_fsst is followed by 3 bytes of padding):
// {u32 total_decompressed_size; bool use_fsst; u32 fsst_offsets_offset; u32 lengths_offset;
// u32 num_codes; u32 codes_offset; bool use_rle_optimized_path; u8 codes_scheme;
// u8 lengths_scheme; u8 data[]}; every *_offset counts from `data`.
// Without FSST `data` starts with (num_codes + 1) u32 slots (offsets from `data`) and the
// dictionary bytes; with FSST it starts with the exported symbol table (SDICT_FSST_HEADER bytes),
// the compressed strings, (num_codes + 1) u32 offsets into them and the lengths stream.
constexpr uint32_t SDICT_TOTAL_SIZE = 0, SDICT_USE_FSST = 4, SDICT_FSST_OFFSETS_OFFSET = 8,
                   SDICT_LENGTHS_OFFSET = 12, SDICT_NUM_CODES = 16, SDICT_CODES_OFFSET = 20,
                   SDICT_USE_RLE = 24, SDICT_CODES_SCHEME = 25, SDICT_LENGTHS_SCHEME = 26,
                   SDICT_DATA = 27;
constexpr uint32_t SDICT_FSST_HEADER = 8 + 1 + 8 + 2048 + 1;  // FSST_MAXHEADER of the fsst library

// Pseudodecimal constants (btrblocks/scheme/double/Pseudodecimal.cpp:16-18)
constexpr uint32_t DEC_MAX_EXPONENT = 22;
constexpr uint32_t DEC_EXCEPTION_CODE = 23;
constexpr uint32_t DEC_INDEX_MASK = 0x1F;
constexpr uint32_t DEC_GROUP_ROWS = 4;  // rows per entry of the CPU-only exception map (Pseudodecimal.hpp:9)

// FastPFor CompositeCodec<SIMDFastPFor<8>, VariableByte> geometry
constexpr uint32_t PFOR_BLOCK = 256, PFOR_PAGE = 65536;
// FastPFor CompositeCodec<FastBinaryPacking<32>, VariableByte> geometry
constexpr uint32_t FBP_MINIBLOCK = 32;      // values per miniblock
constexpr uint32_t FBP_GROUP = 128;         // values per header word (4 miniblocks)
constexpr uint32_t FBP_MAX_GROUP_WORDS = 129;

// CRoaring native serialization (roaring_bitmap_serialize)
constexpr uint8_t ROAR_TAG_ARRAY_U32 = 1;
constexpr uint8_t ROAR_TAG_CONTAINER = 2;
constexpr uint32_t ROAR_COOKIE_RUNS = 12347;
constexpr uint32_t ROAR_COOKIE_NORUNS = 12346;
constexpr uint32_t ROAR_NO_OFFSET_THRESHOLD = 4;
constexpr uint32_t ROAR_ARRAY_MAX_CARD = 4096;

// ---- unaligned little-endian readers
BTR_HD uint32_t rd32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
BTR_HD uint64_t rd64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
BTR_HD uint16_t rd16(const uint8_t* p) {
  uint16_t v;
  memcpy(&v, p, 2);
  return v;
}
inline void wr32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }
inline void wr64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }
inline void wr16(uint8_t* p, uint16_t v) { memcpy(p, &v, 2); }

}  // namespace btrfmt
