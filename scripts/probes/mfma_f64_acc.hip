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
  for (int it = 0; it < (MODE < 3 ? iters : 0); ++it) {
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
  if (MODE == 3 || MODE == 4) {
    // the dense_batch.hip pattern: four conversions of the NEXT unit, then four MFMAs
    float x[4] = {(float)a0 + threadIdx.x, 2.f, 3.f, 4.f};
    double tc[4] = {a[0], a[1], a[2], a[3]}, tn[4];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; i += 4) {
        if (MODE == 3)
          asm volatile(
              "v_cvt_f64_f32 %4, %12\n\tv_cvt_f64_f32 %5, %13\n\tv_cvt_f64_f32 %6, %14\n\t"
              "v_cvt_f64_f32 %7, %15\n\t"
              "v_mfma_f64_16x16x4_f64 %0, %8, %16, %0\n\tv_mfma_f64_16x16x4_f64 %1, %9, %16, %1\n\t"
              "v_mfma_f64_16x16x4_f64 %2, %10, %16, %2\n\tv_mfma_f64_16x16x4_f64 %3, %11, %16, %3"
              : "+v"(acc[i]), "+v"(acc[i + 1]), "+v"(acc[i + 2]), "+v"(acc[i + 3]),
                "=&v"(tn[0]), "=&v"(tn[1]), "=&v"(tn[2]), "=&v"(tn[3])
              : "v"(tc[0]), "v"(tc[1]), "v"(tc[2]), "v"(tc[3]),
                "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(b));
        else
          asm volatile(
              "v_mfma_f64_16x16x4_f64 %0, %8, %16, %0\n\tv_mfma_f64_16x16x4_f64 %1, %9, %16, %1\n\t"
              "v_mfma_f64_16x16x4_f64 %2, %10, %16, %2\n\tv_mfma_f64_16x16x4_f64 %3, %11, %16, %3\n\t"
              "v_mov_b64 %4, %8\n\tv_mov_b64 %5, %9\n\tv_mov_b64 %6, %10\n\tv_mov_b64 %7, %11"
              : "+v"(acc[i]), "+v"(acc[i + 1]), "+v"(acc[i + 2]), "+v"(acc[i + 3]),
                "=&v"(tn[0]), "=&v"(tn[1]), "=&v"(tn[2]), "=&v"(tn[3])
              : "v"(tc[0]), "v"(tc[1]), "v"(tc[2]), "v"(tc[3]),
                "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(b));
#pragma unroll
        for (int e = 0; e < 4; ++e) tc[e] = tn[e];
      }
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
         MODE == 1 ? "asm, accumulators in AGPRs" : MODE == 2 ? "asm, accumulators in VGPRs" : MODE == 3 ? "asm units: 4 f32->f64 conversions + 4 MFMAs, VGPR accumulators" : MODE == 4 ? "asm units: 4 MFMAs + 4 v_mov, VGPR accumulators" : "builtin", ms, ms * 1e6 / per_simd);
}
int main() {
  double* d; hipMalloc(&d, 64);
  run<8, 0, 1>(d); run<8, 0, 4>(d); run<16, 0, 1>(d); run<16, 0, 4>(d);
  run<8, 1, 1>(d); run<8, 1, 4>(d); run<16, 1, 1>(d); run<16, 1, 4>(d); run<32, 1, 4>(d);
  run<8, 2, 1>(d); run<8, 2, 4>(d); run<16, 2, 4>(d);
  run<16, 3, 4>(d); run<16, 4, 4>(d);
  return 0;
}
