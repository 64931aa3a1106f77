// Streaming-read patterns on MI355X: how much does the ADDRESS ORDER of a
// read-only stream matter?  Build: hipcc -O3 --offload-arch=gfx950 -o hbm_patterns hbm_patterns.hip
//   mode 0: grid-stride over the whole buffer (all CUs sweep one window)
//   mode 1: one contiguous chunk per workgroup, one contiguous sub-chunk per wave
//   mode 2: one contiguous chunk per workgroup, waves interleaved at 1 KB
//   mode 3: like 1 but sub-chunks of 32 KB dealt round-robin to the waves of a WG
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void rd(const v4u* __restrict__ src,
                                           int64_t n16, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v4u a = {0, 0, 0, 0};
  if (MODE == 0) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
      v4u x0 = __builtin_nontemporal_load(src + i);
      v4u x1 = __builtin_nontemporal_load(src + i + stride);
      v4u x2 = __builtin_nontemporal_load(src + i + 2 * stride);
      v4u x3 = __builtin_nontemporal_load(src + i + 3 * stride);
      a ^= x0 ^ x1 ^ x2 ^ x3;
    }
  } else {
    const int64_t per_wg = n16 / gridDim.x;          // 16-B units
    const v4u* base = src + per_wg * blockIdx.x;
    if (MODE == 1) {
      const int64_t per_wave = per_wg / nw;
      const v4u* p = base + per_wave * wave + lane;
      for (int64_t i = 0; i + 255 < per_wave; i += 256) {
        v4u x0 = __builtin_nontemporal_load(p + i);
        v4u x1 = __builtin_nontemporal_load(p + i + 64);
        v4u x2 = __builtin_nontemporal_load(p + i + 128);
        v4u x3 = __builtin_nontemporal_load(p + i + 192);
        a ^= x0 ^ x1 ^ x2 ^ x3;
      }
    } else if (MODE == 2) {
      const v4u* p = base + wave * 64 + lane;
      const int64_t st = (int64_t)nw * 64;
      for (int64_t i = 0; i + 3 * st < per_wg; i += 4 * st) {
        v4u x0 = __builtin_nontemporal_load(p + i);
        v4u x1 = __builtin_nontemporal_load(p + i + st);
        v4u x2 = __builtin_nontemporal_load(p + i + 2 * st);
        v4u x3 = __builtin_nontemporal_load(p + i + 3 * st);
        a ^= x0 ^ x1 ^ x2 ^ x3;
      }
    } else {
      const int64_t chunk = 2048;  // 32 KB in 16-B units
      for (int64_t c = wave * chunk; c + chunk <= per_wg; c += nw * chunk) {
        const v4u* p = base + c + lane;
        for (int64_t i = 0; i < chunk; i += 256) {
          v4u x0 = __builtin_nontemporal_load(p + i);
          v4u x1 = __builtin_nontemporal_load(p + i + 64);
          v4u x2 = __builtin_nontemporal_load(p + i + 128);
          v4u x3 = __builtin_nontemporal_load(p + i + 192);
          a ^= x0 ^ x1 ^ x2 ^ x3;
        }
      }
    }
  }
  const unsigned f = a.x ^ a.y ^ a.z ^ a.w;
  if (f == 0x9E3779B9u) sink[0] = f;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int64_t bytes = argc > 1 ? atoll(argv[1]) : 240000000;
  const int reps = 50;
  const int64_t n16 = bytes / 16;
  void *src, *sink;
  CK(hipMalloc(&src, n16 * 16)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(src, 1, n16 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int threads : {256, 1024}) for (int wgs : {256, 512, 2048}) for (int mode = 0; mode < 4; ++mode) {
    float ms;
    for (int it = -3; it < reps; ++it) {
      if (it == 0) CK(hipEventRecord(e0, 0));
      switch (mode) {
        case 0: hipLaunchKernelGGL(rd<0>, dim3(wgs), dim3(threads), 0, 0, (const v4u*)src, n16, (unsigned*)sink); break;
        case 1: hipLaunchKernelGGL(rd<1>, dim3(wgs), dim3(threads), 0, 0, (const v4u*)src, n16, (unsigned*)sink); break;
        case 2: hipLaunchKernelGGL(rd<2>, dim3(wgs), dim3(threads), 0, 0, (const v4u*)src, n16, (unsigned*)sink); break;
        default: hipLaunchKernelGGL(rd<3>, dim3(wgs), dim3(threads), 0, 0, (const v4u*)src, n16, (unsigned*)sink); break;
      }
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("threads %4d wgs %4d mode %d: %.1f us/launch  %.0f GB/s\n", threads, wgs, mode, 1e3 * ms / reps, n16 * 16.0 * reps / ms / 1e6);
  }
  return 0;
}
