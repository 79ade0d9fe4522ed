// Host-side (CPU) encoders for the two third-party payload formats BtrBlocks embeds:
// FastPFor's CompositeCodec<FastBinaryPacking<32>, VariableByte> and
// CompositeCodec<SIMDFastPFor<8>, VariableByte> (used through
// btrblocks/extern/FastPFOR.cpp:19-31, 40-53) and CRoaring's native bitmap
// serialization (Roaring::write(buf, false), used in
// btrblocks/scheme/templated/Frequency.hpp:82-85, btrblocks/scheme/double/Pseudodecimal.cpp:178-180
// and btrblocks/extern/RoaringBitmap.cpp:148-156).
//
// The libraries themselves (lemire/FastPFor @ bc375056, RoaringBitmap/CRoaring @ b88b0024)
// are not part of the reference tree; these encoders restate their published layouts.
// Compression stays on the CPU by design: only the decode cascade runs on the GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace btrhost {

// FastBinaryPacking<32> over the first n/32*32 values, VariableByte over the rest.
// Appends 32-bit words to `out`; returns the number of words appended.
size_t fbp_encode(const uint32_t* in, size_t n, std::vector<uint32_t>& out);

// SIMDFastPFor<8> over the first n/256*256 values, VariableByte over the rest. The codec pads to
// 16-byte boundaries of the absolute address: `first_word_mod4` is (address of the first word
// appended / 4) % 4 at the place the stream will finally live.
size_t pfor_encode(const uint32_t* in, size_t n, std::vector<uint32_t>& out, size_t first_word_mod4);

// Serialize a sorted, duplicate-free list of row ids the way roaring_bitmap_serialize does
// after runOptimize(). Appends bytes to `out`; returns the number of bytes appended.
size_t roaring_serialize(const uint32_t* sorted_ids, size_t n, std::vector<uint8_t>& out);

}  // namespace btrhost
