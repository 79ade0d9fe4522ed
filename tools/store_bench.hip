// Store-rate microbenchmark: which store policy and workgroup geometry reaches the fill rate of the
// memory system (hipcc --offload-arch=gfx950 -O3 tools/store_bench.hip -o build/store_bench).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one workgroup per 256 KiB block, THREADS threads, 16 bytes per lane and store, UNROLL stores per iteration
template <int THREADS, int UNROLL, bool NT>
__global__ void __launch_bounds__(THREADS) fill_block(u32* __restrict__ dst, u32 v) {
  u32x4* p = reinterpret_cast<u32x4*>(dst + (size_t)blockIdx.x * 65536);
  const u32x4 val = {v, v, v, v};
  for (u32 base = 0; base < 16384; base += THREADS * UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if (NT) __builtin_nontemporal_store(val, p + base + u * THREADS + threadIdx.x);
      else p[base + u * THREADS + threadIdx.x] = val;
    }
  }
}
// fill_block with at most KEEP stores of a wave in flight behind every store, and per-wave contiguous quarters (QUARTERS)
template <int KEEP, bool QUARTERS, bool PAIR>
__global__ void __launch_bounds__(256) fill_block_paced(u32* __restrict__ dst, u32 v) {
  u32x4* p = reinterpret_cast<u32x4*>(dst + (size_t)blockIdx.x * 65536);
  const u32x4 val = {v, v, v, v};
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (u32 it = 0; it < 64; it++) {  // 64 stores of 16 bytes per lane
    u32 q;
    if (PAIR) q = QUARTERS ? wave * 4096 + (it >> 1) * 128 + lane * 2 + (it & 1) : (it >> 1) * 512 + threadIdx.x * 2 + (it & 1);
    else q = QUARTERS ? wave * 4096 + it * 64 + lane : it * 256 + threadIdx.x;
    p[q] = val;
    if (KEEP < 64) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
  }
}
// as fill_block, but workgroup b starts ROT(b) iterations into its block and wraps around
template <int THREADS, int UNROLL, bool NT>
__global__ void __launch_bounds__(THREADS) fill_block_rot(u32* __restrict__ dst, u32 v, u32 mul) {
  u32x4* p = reinterpret_cast<u32x4*>(dst + (size_t)blockIdx.x * 65536);
  const u32x4 val = {v, v, v, v};
  constexpr u32 kIters = 16384 / (THREADS * UNROLL);
  const u32 rot = (blockIdx.x * mul) % kIters;
  for (u32 it = 0; it < kIters; it++) {
    const u32 base = ((it + rot) % kIters) * THREADS * UNROLL;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if (NT) __builtin_nontemporal_store(val, p + base + u * THREADS + threadIdx.x);
      else p[base + u * THREADS + threadIdx.x] = val;
    }
  }
}
// small workgroups: each writes `per` consecutive 16-byte quads per thread-slot (4 KiB x per / ... )
template <int PER, bool NT>
__global__ void __launch_bounds__(256) fill_small(u32* __restrict__ dst, u32 v) {
  u32x4* p = reinterpret_cast<u32x4*>(dst) + (size_t)blockIdx.x * 256 * PER;
  const u32x4 val = {v, v, v, v};
#pragma unroll
  for (int u = 0; u < PER; u++) {
    if (NT) __builtin_nontemporal_store(val, p + u * 256 + threadIdx.x); else p[u * 256 + threadIdx.x] = val;
  }
}
// workgroups of THREADS threads, STORES 16-byte stores per thread, the workgroup's bytes contiguous
template <int THREADS, int STORES>
__global__ void __launch_bounds__(THREADS) fill_wg(u32* __restrict__ dst, u32 v) {
  u32x4* p = reinterpret_cast<u32x4*>(dst) + (size_t)blockIdx.x * THREADS * STORES;
  const u32x4 val = {v, v, v, v};
#pragma unroll
  for (int u = 0; u < STORES; u++) p[u * THREADS + threadIdx.x] = val;
}
// every thread one 16-byte store per grid stride (torch-like elementwise)
template <bool NT>
__global__ void __launch_bounds__(256) fill_flat(u32* __restrict__ dst, u32 v, size_t quads) {
  u32x4* p = reinterpret_cast<u32x4*>(dst);
  const u32x4 val = {v, v, v, v};
  const size_t base = (size_t)blockIdx.x * 256 * 4;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const size_t i = base + u * 256 + threadIdx.x;
    if (i < quads) { if (NT) __builtin_nontemporal_store(val, p + i); else p[i] = val; }
  }
}
// copies: one workgroup per 256 KiB block (as the engine's UNCOMPRESSED root) and one 4 KiB page per workgroup
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) copy_block(u32* __restrict__ dst, const u32* __restrict__ src) {
  u32x4* p = reinterpret_cast<u32x4*>(dst + (size_t)blockIdx.x * 65536);
  const u32x4* q = reinterpret_cast<const u32x4*>(src + (size_t)blockIdx.x * 65536);
  for (u32 base = 0; base < 16384; base += 256 * UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = q[base + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      if (NT) __builtin_nontemporal_store(v[u], p + base + u * 256 + threadIdx.x);
      else p[base + u * 256 + threadIdx.x] = v[u];
    }
  }
}
template <bool NT>
__global__ void __launch_bounds__(256) copy_page(u32* __restrict__ dst, const u32* __restrict__ src) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const u32x4 v = reinterpret_cast<const u32x4*>(src)[i];
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst) + i);
  else reinterpret_cast<u32x4*>(dst)[i] = v;
}
template <typename F>
static void timeit(const char* name, F launch, size_t bytes) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < 3; i++) launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int i = 0; i < 20; i++) launch();
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  printf("%-44s %8.1f GB/s  %.4f ms\n", name, bytes * 20.0 / (ms * 1e-3) / 1e9, ms / 20);
}
int main() {
  const size_t n = 268435456;  // 1 GiB of u32
  u32* d; CHECK(hipMalloc(&d, n * 4));
  const u32 blocks = n / 65536;
  timeit("block 256thr unroll2 nt (engine fill)", [&] { fill_block<256, 2, true><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("block 256thr unroll2 plain", [&] { fill_block<256, 2, false><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("block 256thr unroll8 nt", [&] { fill_block<256, 8, true><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("block 256thr unroll8 plain", [&] { fill_block<256, 8, false><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("block 512thr unroll2 nt", [&] { fill_block<512, 2, true><<<blocks, 512>>>(d, 7); }, n * 4);
  timeit("block 512thr unroll2 plain", [&] { fill_block<512, 2, false><<<blocks, 512>>>(d, 7); }, n * 4);
  timeit("block 1024thr unroll2 plain", [&] { fill_block<1024, 2, false><<<blocks, 1024>>>(d, 7); }, n * 4);
  for (u32 mul : {1u, 3u, 7u, 13u}) {
    char name[64]; snprintf(name, sizeof name, "block 256thr unroll2 rotated x%u nt", mul);
    timeit(name, [&] { fill_block_rot<256, 2, true><<<blocks, 256>>>(d, 7, mul); }, n * 4);
    snprintf(name, sizeof name, "block 256thr unroll2 rotated x%u plain", mul);
    timeit(name, [&] { fill_block_rot<256, 2, false><<<blocks, 256>>>(d, 7, mul); }, n * 4);
  }
  timeit("paced keep 0", [&] { fill_block_paced<0, false, false><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("paced keep 1", [&] { fill_block_paced<1, false, false><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("paced keep 2", [&] { fill_block_paced<2, false, false><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("paced keep 4", [&] { fill_block_paced<4, false, false><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("paced keep 8", [&] { fill_block_paced<8, false, false><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("paced keep 16", [&] { fill_block_paced<16, false, false><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("unpaced, waves interleaved", [&] { fill_block_paced<64, false, false><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("unpaced, wave quarters", [&] { fill_block_paced<64, true, false><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("keep 2, wave quarters", [&] { fill_block_paced<2, true, false><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("unpaced, 32 B per lane, interleaved", [&] { fill_block_paced<64, false, true><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("unpaced, 32 B per lane, quarters", [&] { fill_block_paced<64, true, true><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("keep 2, 32 B per lane, interleaved", [&] { fill_block_paced<2, false, true><<<blocks, 256>>>(d, 7); }, n * 4);
  timeit("wg 64 thr x 1 store (1 KiB)", [&] { fill_wg<64, 1><<<n / 256, 64>>>(d, 7); }, n * 4);
  timeit("wg 64 thr x 4 stores (4 KiB)", [&] { fill_wg<64, 4><<<n / 1024, 64>>>(d, 7); }, n * 4);
  timeit("wg 128 thr x 2 stores (4 KiB)", [&] { fill_wg<128, 2><<<n / 1024, 128>>>(d, 7); }, n * 4);
  timeit("wg 512 thr x 1 store (8 KiB)", [&] { fill_wg<512, 1><<<n / 2048, 512>>>(d, 7); }, n * 4);
  timeit("wg 1024 thr x 1 store (16 KiB)", [&] { fill_wg<1024, 1><<<n / 4096, 1024>>>(d, 7); }, n * 4);
  timeit("wg 1024 thr x 2 stores (32 KiB)", [&] { fill_wg<1024, 2><<<n / 8192, 1024>>>(d, 7); }, n * 4);
  timeit("wg 256 thr x 3 stores (12 KiB)", [&] { fill_wg<256, 3><<<n / 3072, 256>>>(d, 7); }, n * 4);
  timeit("small 4 KiB workgroups plain", [&] { fill_small<1, false><<<n / 1024, 256>>>(d, 7); }, n * 4);
  timeit("small 4 KiB workgroups nt", [&] { fill_small<1, true><<<n / 1024, 256>>>(d, 7); }, n * 4);
  timeit("small 8 KiB workgroups plain", [&] { fill_small<2, false><<<n / 2048, 256>>>(d, 7); }, n * 4);
  timeit("small 16 KiB workgroups plain", [&] { fill_small<4, false><<<n / 4096, 256>>>(d, 7); }, n * 4);
  timeit("small 64 KiB workgroups plain", [&] { fill_small<16, false><<<n / 16384, 256>>>(d, 7); }, n * 4);
  const size_t quads = n / 4;
  const u32 flat = (u32)((quads + 1023) / 1024);
  timeit("flat 16 KiB per workgroup nt", [&] { fill_flat<true><<<flat, 256>>>(d, 7, quads); }, n * 4);
  timeit("flat 16 KiB per workgroup plain", [&] { fill_flat<false><<<flat, 256>>>(d, 7, quads); }, n * 4);
  u32* src; CHECK(hipMalloc(&src, n * 4)); CHECK(hipMemset(src, 1, n * 4));
  timeit("copy block 256 KiB unroll4 nt (read+write)", [&] { copy_block<4, true><<<blocks, 256>>>(d, src); }, n * 8);
  timeit("copy block 256 KiB unroll4 plain", [&] { copy_block<4, false><<<blocks, 256>>>(d, src); }, n * 8);
  timeit("copy block 256 KiB unroll8 nt", [&] { copy_block<8, true><<<blocks, 256>>>(d, src); }, n * 8);
  timeit("copy 4 KiB page per workgroup nt", [&] { copy_page<true><<<n / 1024, 256>>>(d, src); }, n * 8);
  timeit("copy 4 KiB page per workgroup plain", [&] { copy_page<false><<<n / 1024, 256>>>(d, src); }, n * 8);
  timeit("hipMemcpyAsync device to device", [&] { CHECK(hipMemcpyAsync(d, src, n * 4, hipMemcpyDeviceToDevice, 0)); }, n * 8);
  CHECK(hipMemset(d, 0, n * 4));
  timeit("hipMemsetAsync", [&] { CHECK(hipMemsetAsync(d, 1, n * 4, 0)); }, n * 4);
  return 0;
}
