// How fast can ONE wave fill LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB
// per instruction) from an L2-resident vector, and what does rewriting M0
// between the instructions cost?
//   mode 0: M0 saved / set / restored around every DMA (the guide's glds16 recipe)
//   mode 1: M0 set before every DMA, never read or restored
//   mode 2: M0 set once per 4 DMAs, the instruction's offset field (0, 1024,
//           2048, 3072) advances both addresses
// Build: hipcc -O3 --offload-arch=gfx950 -o lds_dma_rate lds_dma_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k(const double* __restrict__ x, int n_chunk,
                                          int reps, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave != 15) return;
  const unsigned base = (unsigned)(uintptr_t)lds;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    for (int ch = 0; ch < n_chunk; ch += 4) {
      const double* src = x + (size_t)ch * 128 + 2 * lane;
      const unsigned dst = __builtin_amdgcn_readfirstlane(base + ch * 1024);
      if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                       "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(src + u * 128), "s"(dst + u * 1024) : "memory");
        }
      } else if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                       : : "v"(src + u * 128), "s"(dst + u * 1024) : "memory");
      } else {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off\n\t"
                     "global_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\t"
                     "global_load_lds_dwordx4 %0, off offset:3072"
                     : : "v"(src), "s"(dst) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    out[blockIdx.x * 2] = t1 - t0;
    // checksum of what landed (mode 2 must deliver the same image)
    double s = 0.;
    for (int i = 0; i < n_chunk * 128; ++i) s += lds[i];
    out[blockIdx.x * 2 + 1] = (unsigned long long)s;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int n_chunk = 56, reps = 200;  // 56 KiB per fill
  double* x;
  unsigned long long* out;
  CK(hipMalloc(&x, 8 * 128 * 64));
  CK(hipMalloc(&out, 16 * 256));
  double* hx = (double*)malloc(8 * 128 * 64);
  for (int i = 0; i < 128 * 64; ++i) hx[i] = i % 7;
  CK(hipMemcpy(x, hx, 8 * 128 * 64, hipMemcpyHostToDevice));
  auto run = [&](auto kern, const char* name) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int wgs : {1, 256}) {
      unsigned long long h[512];
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      float ms = 0.f;
      for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(1024), 64 * 1024, 0, x, n_chunk, reps, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(h, out, 16 * wgs, hipMemcpyDeviceToHost);
      double mean = 0;
      for (int b = 0; b < wgs; ++b) mean += (double)h[2 * b];
      mean /= wgs;
      // (the launch itself is a few us of the event interval; reps = 200)
      const double us = 1e3 * ms / reps;
      printf("%s  workgroups %3d: %.0f s_memtime ticks, %.2f us per 56 KiB fill = %.1f GB/s per loader wave, checksum %llu\n",
             name, wgs, mean / reps, us, 56 * 1024 / us / 1e3, h[1]);
    }
  };
  run(k<0>, "mode 0 (save/set/restore M0)");
  run(k<1>, "mode 1 (set M0 per DMA)     ");
  run(k<2>, "mode 2 (M0 per 4, offsets)  ");
  return 0;
}
