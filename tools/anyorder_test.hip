// Do two kernels of one stream overlap when the second is launched with hipExtAnyOrderLaunch?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)
__global__ void spin(unsigned long long ticks, unsigned* out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
int main() {
  unsigned* d; CHECK(hipMalloc(&d, 4096));
  hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      CHECK(hipEventRecord(a, s));
      // 128 workgroups of one wave each for 200 us (20000 ticks of the 100 MHz clock): far from filling the GPU
      hipExtLaunchKernelGGL(spin, dim3(128), dim3(64), 0, s, nullptr, nullptr, 0, 20000ull, d);
      hipExtLaunchKernelGGL(spin, dim3(128), dim3(64), 0, s, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, 20000ull, d + 128);
      CHECK(hipGetLastError());
      CHECK(hipEventRecord(b, s)); CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      printf("second launch %s: %.3f ms for two 0.2 ms kernels\n", mode ? "any-order" : "in order ", ms);
    }
  }
  return 0;
}
