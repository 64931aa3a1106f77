// How far apart do the 16 waves of a 1024-thread workgroup START on MI355X?
// (s_memtime at kernel entry per wave, spread inside each workgroup.)
// Build: hipcc -O3 --offload-arch=gfx950 -o launch_stagger launch_stagger.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

template <int MODE>
__global__ __launch_bounds__(1024, 4) void k(unsigned long long* out, const int* chain, int lds_words) {
  extern __shared__ double lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int wave = threadIdx.x >> 6;
  unsigned long long t1 = t0;
  if (MODE >= 1) {  // zero the accumulators like the real prologue
    for (int r = threadIdx.x; r < lds_words; r += 1024) lds[r] = 0.;
  }
  if (MODE >= 2) {  // two dependent loads like wave_desc -> descs
    int a = chain[blockIdx.x * 16 + wave];
    int b = chain[4096 + (a & 4095) + (threadIdx.x & 63)];
    if (b == 123456789) lds[0] = 1.;
    t1 = __builtin_amdgcn_s_memtime();
  }
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 16 + wave) * 2] = t0;
    out[(blockIdx.x * 16 + wave) * 2 + 1] = t1;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int wgs = 253;
  unsigned long long* d; int* chain;
  CK(hipMalloc(&d, wgs * 16 * 2 * 8));
  CK(hipMalloc(&chain, 16384 * 4));
  CK(hipMemset(chain, 0, 16384 * 4));
  std::vector<unsigned long long> h(wgs * 16 * 2);
  for (int mode = 0; mode < 3; ++mode) for (int ldskb : {8, 150}) {
    const int lds_words = ldskb * 1024 / 8;
    for (int rep = 0; rep < 3; ++rep) {
      if (mode == 0) { CK(hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(1024), ldskb * 1024, 0, d, chain, lds_words); }
      if (mode == 1) { CK(hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(1024), ldskb * 1024, 0, d, chain, lds_words); }
      if (mode == 2) { CK(hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(1024), ldskb * 1024, 0, d, chain, lds_words); }
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    double spread = 0, spread1 = 0; double byw[16] = {0};
    for (int w = 0; w < wgs; ++w) {
      unsigned long long lo = ~0ull, hi = 0, hi1 = 0;
      for (int i = 0; i < 16; ++i) { lo = std::min(lo, h[(w * 16 + i) * 2]); hi = std::max(hi, h[(w * 16 + i) * 2]); hi1 = std::max(hi1, h[(w * 16 + i) * 2 + 1]); }
      spread += (double)(hi - lo); spread1 += (double)(hi1 - lo);
      for (int i = 0; i < 16; ++i) byw[i] += (double)(h[(w * 16 + i) * 2] - lo);
    }
    printf("mode %d lds %3d KB: mean spread of wave start inside a workgroup %.0f ticks; last wave past its loads %.0f ticks after the first start; start offset by wave:", mode, ldskb, spread / wgs, spread1 / wgs);
    for (int i = 0; i < 16; ++i) printf(" %.0f", byw[i] / wgs);
    printf("\n");
  }
  return 0;
}
