// Round trip of a two-column relation through the C++ interface: compress on the CPU, decode on
// the GPU, compare. Follows the flow of the reference's tools/examples/compression.cpp:56-117
// (relation -> ranges -> chunk -> Datablock::compress -> Datablock::decompress -> compare).
//   usage: compression [rows] [scheme spec]
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <random>

#include "btrblocks_amd.hpp"

using namespace btrblocks_amd;

// `unique` distinct values in runs of up to `runlength`
template <typename T>
static std::vector<T> generateData(size_t size, size_t unique, size_t runlength, int seed) {
  std::vector<T> data(size);
  std::mt19937 gen(seed);
  for (size_t i = 0; i < size;) {
    const T v = static_cast<T>(gen() % unique) / (sizeof(T) == 8 ? 100.0 : 1);
    const size_t run = 1 + gen() % runlength;
    for (size_t j = 0; j < run && i < size; j++, i++) data[i] = v;
  }
  return data;
}

int main(int argc, char** argv) {
  const size_t size = argc > 1 ? strtoull(argv[1], nullptr, 10) : 640000;
  const std::string spec = argc > 2 ? argv[2] : "auto";
  try {
    Relation to_compress;
    to_compress.addColumn({"ints", generateData<int32_t>(size, 1 << 12, 40, 42)});
    to_compress.addColumn({"dbls", generateData<double>(size, 1 << 10, 20, 69)});

    // one block per 65536 rows, as the reference's tools do with Relation::getRanges
    const std::vector<Range> ranges = to_compress.getRanges(SplitStrategy::SEQUENTIAL, 65536);
    Datablock compressor(to_compress, 0, spec);
    size_t compressed = 0, raw = 0;
    bool check = true;
    for (size_t chunk_i = 0; chunk_i < ranges.size(); chunk_i++) {
      Chunk input = to_compress.getChunk(ranges, chunk_i);
      BytesArray output;
      const OutputBlockStats stats = compressor.compress(input, output);
      compressed += stats.total_data_size;
      raw += stats.total_uncompressed;
      Chunk decompressed = compressor.decompress(output);
      check = check && (input == decompressed);
    }
    std::cout << "relation of " << ranges.size() << " blocks: " << raw << " raw bytes -> " << compressed << " compressed bytes ("
              << (compressed ? (double)raw / compressed : 0) << " x)" << std::endl;
    std::cout << (check ? "block round trip: every block decodes to its input" : "block round trip: MISMATCH")
              << std::endl;

    // The column-part path the reference's decompression-speed tool times
    // (tools/conversion/decompression-speed.cpp:40-64): one BtrReader per part image, readColumn per
    // chunk, getBitmap / releaseBitmap per chunk. Every 7th row of the integer column is null.
    {
      const std::vector<int32_t> ints = generateData<int32_t>(size, 1 << 12, 40, 42);
      std::vector<uint8_t> valid(size, 1);
      for (size_t i = 0; i < size; i += 7) valid[i] = 0;
      std::vector<uint8_t> image(btr_compress_column_bound(0, size, 65536));
      size_t image_bytes = 0;
      btrblocks_amd::check(btr_compress_column(0, ints.data(), valid.data(), size, 65536, spec.c_str(), nullptr, 0, image.data(),
                                       image.size(), &image_bytes));
      BtrReader reader(image.data(), image_bytes);
      bool part_ok = reader.getChunkCount() == (size + 65535) / 65536;
      std::vector<u8> chunk;
      size_t row = 0;
      for (u32 i = 0; i < reader.getChunkCount(); i++) {
        const u32 n = reader.getTupleCount(i);
        part_ok = part_ok && reader.getDecompressedDataSize(i) == (size_t)n * 4 && !reader.readColumn(chunk, i);
        const BITMAP* bm = reader.getBitmap(i);
        const int32_t* got = reinterpret_cast<const int32_t*>(chunk.data());
        for (u32 r = 0; r < n && part_ok; r++)
          part_ok = bm[r] == valid[row + r] && (!valid[row + r] || got[r] == ints[row + r]);
        reader.releaseBitmap(i);
        row += n;
      }
      std::cout << (part_ok ? "part reader round trip: every chunk decodes to its input" : "part reader round trip: MISMATCH")
                << std::endl;
      check = check && part_ok;
    }
    return check ? 0 : 1;
  } catch (const Generic_Exception& e) {
    std::cerr << "btrblocks_amd error " << e.status << ": " << e.what() << std::endl;
    return 2;
  }
}
