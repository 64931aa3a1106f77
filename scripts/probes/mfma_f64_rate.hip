// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: W waves per CU, each running
// a loop of independent MFMAs (NACC accumulators).  Prints cycles per MFMA per
// SIMD and TFLOP/s chip-wide.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void probe(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0., 0., 0., 0.};
  double a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  d4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s[0] == 12345.678) out[0] = s[1];
}
template <int NACC>
void run(int waves_per_cu, double* d) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = 64 * waves_per_cu;
  probe<NACC><<<256, threads>>>(d, 10, 1., 2.);
  hipEventRecord(e0);
  probe<NACC><<<256, threads>>>(d, iters, 1., 2.);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)iters * 8 * NACC * waves_per_cu * 256;
  const double tflops = n_mfma * 2048 / (ms * 1e-3) / 1e12;
  printf("NACC=%d waves/CU=%2d: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n",
         NACC, waves_per_cu, ms, tflops,
         ms * 1e6 / (n_mfma / 1024.));
}
int main() {
  double* d; hipMalloc(&d, 64);
  for (int w : {4, 8, 16}) { run<1>(w, d); run<2>(w, d); run<4>(w, d); run<8>(w, d); }
  return 0;
}
