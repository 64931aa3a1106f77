// How many bytes must a CU keep in flight to stream at the HBM rate?
// One 1024-thread workgroup per CU (the tiled kernel's shape) or two; every
// wave keeps D 16-byte-per-lane loads (D KiB) in flight over its own contiguous
// region, like the tiled kernel's register ring.
// Build: hipcc -O3 --offload-arch=gfx950 -o inflight inflight.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(1024) void rd(const v4u* __restrict__ src,
                                           int64_t n16, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t per_wg = n16 / gridDim.x;
  const int64_t per_wave = per_wg / nw;
  const v4u* p = src + per_wg * blockIdx.x + per_wave * wave + lane;
  v4u acc = {0, 0, 0, 0};
  v4u r[D];
  const int64_t steps = per_wave / 64;  // 1 KiB wave loads
#pragma unroll
  for (int k = 0; k < D; ++k) r[k] = __builtin_nontemporal_load(p + (int64_t)k * 64);
  int64_t i = D;
  for (; i + D <= steps; i += D) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      acc ^= r[k];  // waits for the oldest load only
      r[k] = __builtin_nontemporal_load(p + (i + k) * 64);
    }
  }
#pragma unroll
  for (int k = 0; k < D; ++k) acc ^= r[k];
  const unsigned f = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (f == 0x9E3779B9u) sink[0] = f;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int D>
static int run(const void* src, int64_t n16, void* sink, int wgs, hipEvent_t e0, hipEvent_t e1) {
  const int reps = 30;
  float ms;
  for (int it = -3; it < reps; ++it) {
    if (it == 0) CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rd<D>, dim3(wgs), dim3(1024), 0, 0, (const v4u*)src, n16, (unsigned*)sink);
  }
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  wgs %4d  depth %2d KiB/wave: %.1f us  %.0f GB/s\n", wgs, D, 1e3 * ms / reps, n16 * 16.0 * reps / ms / 1e6);
  return 0;
}

int main(int argc, char** argv) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  void* sink;
  CK(hipMalloc(&sink, 64));
  for (int64_t bytes : {(int64_t)233000000, (int64_t)2000000000}) {
    const int64_t n16 = bytes / 16;
    void* src;
    CK(hipMalloc(&src, n16 * 16));
    CK(hipMemset(src, 1, n16 * 16));
    printf("== %lld bytes\n", (long long)bytes);
    for (int wgs : {256, 512}) {
      run<1>(src, n16, sink, wgs, e0, e1);
      run<2>(src, n16, sink, wgs, e0, e1);
      run<3>(src, n16, sink, wgs, e0, e1);
      run<4>(src, n16, sink, wgs, e0, e1);
      run<6>(src, n16, sink, wgs, e0, e1);
      run<8>(src, n16, sink, wgs, e0, e1);
      run<12>(src, n16, sink, wgs, e0, e1);
      run<16>(src, n16, sink, wgs, e0, e1);
    }
    CK(hipFree(src));
  }
  return 0;
}
