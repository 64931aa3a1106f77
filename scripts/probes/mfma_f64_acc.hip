// v_mfma_f64_16x16x4_f64 issue rate against the NUMBER of accumulators in rotation and
// where they live (AGPRs through inline asm "+a", as dense_batch.hip does), with the A
// operand changing every MFMA.  4 waves per CU (one per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_acc mfma_f64_acc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, int MODE, int NA>
__global__ __launch_bounds__(256) void probe(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0., 0., 0., 0.};
  double a[4] = {a0 + threadIdx.x, a0 + 1., a0 + 2., a0 + 3.};
  double b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MODE == 0)
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & (NA - 1)], b, acc[i], 0, 0, 0);
      else if (MODE == 1)
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i & (NA - 1)]), "v"(b));
      else
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & (NA - 1)]), "v"(b));
    }
  }
  d4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s[0] == 12345.678) out[0] = s[1];
}
template <int NACC, int MODE, int NA>
void run(double* d) {
  const int iters = 4000 / NACC * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) probe<NACC, MODE, NA><<<256, 256>>>(d, iters, 1., 2.);
  hipEventRecord(e0);
  probe<NACC, MODE, NA><<<256, 256>>>(d, iters, 1., 2.);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = (double)iters * NACC;
  printf("NACC=%2d, %d A registers, %s: %.3f ms, %.1f ns per MFMA per SIMD\n", NACC, NA,
         MODE == 1 ? "asm, accumulators in AGPRs" : MODE == 2 ? "asm, accumulators in VGPRs" : "builtin", ms, ms * 1e6 / per_simd);
}
int main() {
  double* d; hipMalloc(&d, 64);
  run<8, 0, 1>(d); run<8, 0, 4>(d); run<16, 0, 1>(d); run<16, 0, 4>(d);
  run<8, 1, 1>(d); run<8, 1, 4>(d); run<16, 1, 1>(d); run<16, 1, 4>(d); run<32, 1, 4>(d);
  run<8, 2, 1>(d); run<8, 2, 4>(d); run<16, 2, 4>(d);
  return 0;
}
