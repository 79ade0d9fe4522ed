// Host-side plan compiler: walks the header tree of one compressed column (the same walk as the
// reference's fullDescription(), e.g. btrblocks/scheme/templated/RLE.hpp:87-101), validates
// every offset against the byte length the caller supplied, and emits the flat program that the
// decode kernel interprets (plan.h).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "plan.h"

namespace btrplan {

struct PlanError : std::runtime_error {
  int status;  // btr_status value
  PlanError(int st, const std::string& msg) : std::runtime_error(msg), status(st) {}
};

struct ColumnSource {
  const uint8_t* host;    // host copy of the bytes the scheme payload lives in
  size_t bytes;           // length of that buffer
  uint64_t dev_addr;      // device address the same bytes are resident at
  size_t payload_off;     // offset of the root scheme payload inside the buffer
  size_t nullmap_off;     // offset of the serialized nullmap inside the buffer
  uint8_t column_type;    // btrfmt::COL_INTEGER / COL_DOUBLE
  uint8_t scheme;         // root scheme code
  uint8_t nullmap_type;   // btrfmt::NM_*
  uint32_t tuple_count;
  bool want_nullmap;      // emit the byte-per-row nullmap
};

struct BuiltProgram {
  std::vector<uint8_t> program;  // ProgramHeader | Step[] | Pipe[] | Seq[]
  uint32_t lds_bytes = 0;        // dynamic LDS the workgroup needs (program copy + tables)
  uint32_t scratch_bytes = 0;    // HBM scratch slice for tables that did not fit in LDS
  uint64_t needed_bytes = 0;     // compressed bytes the decoder actually reads
  uint32_t depth = 0;            // cascade depth of the tree
  uint32_t kernel = 0;           // KERNEL_* that decodes this program
  std::string description;       // scheme tree in text form, e.g. "RLE(BP,BP)"
};

// Throws PlanError. Envelope parsing (ColumnChunkMeta / ColumnMeta) is done by the caller.
void build_program(const ColumnSource& src, BuiltProgram& out);

}  // namespace btrplan
