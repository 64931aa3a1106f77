// Where does a Polya-Gamma draw spend its time on the MI355X?  The pieces of
// PolyaGamma::jacobi (csrc/samplers.hpp) timed one by one over n = 2^20
// elements, one lane per element, z = |psi| / 2 as given on the command line:
//   0 philox   one Philox block (10 rounds) and two uniforms
//   1 weights  right_mass(z, rate): two log Phi (erfc + log), three exp, logs
//   2 series   series_accept(x = .3): one uniform, two or three series terms
//   3 ig-loop  the truncated inverse-Gaussian rejection loop (sequential)
//   4 exp      the exponential-piece proposal (one log)
//   5 whole    jacobi(): the draw as the kernel of rounds 1-4 did it
//   6 ig-one   ONE inverse-Gaussian attempt (no loop)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../bayes-bridge_amd/csrc
//        -o pg_parts pg_parts.hip ;  ./pg_parts [z]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "philox.hpp"
#include "samplers.hpp"

using namespace bbx;

template <int MODE>
__global__ __launch_bounds__(256) void k(int64_t n, double z, uint64_t seed,
                                         double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256) {
    Philox g(seed, 3, (uint64_t)i);
    const double zz = z * (1. + 1e-9 * (double)(i & 1023));   // defeat hoisting
    const double rate = 0.5 * zz * zz + 0.125 * kPi * kPi;
    double r = 0.;
    if (MODE == 0) r = g.uniform() + g.uniform();
    if (MODE == 1) r = PolyaGamma::right_mass(zz, rate);
    if (MODE == 2) r = PolyaGamma::series_accept(g, .3 + 1e-6 * zz) ? 1. : 0.;
    if (MODE == 3) r = PolyaGamma::trunc_inv_gauss(g, zz, PolyaGamma::kCut);
    if (MODE == 4) r = PolyaGamma::trunc_exp(g, 1. / rate, PolyaGamma::kCut);
    if (MODE == 5) r = PolyaGamma::jacobi(g, zz);
    if (MODE == 6) {
      double x = 0.;
      r = PolyaGamma::trunc_inv_gauss_attempt(g, zz, PolyaGamma::kCut, x) ? x : -x;
    }
    out[i] = r;
  }
}

template <int MODE>
static float run(int64_t n, double z, double* d_out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<MODE><<<2048, 256>>>(n, z, 7, d_out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 10; ++r) k<MODE><<<2048, 256>>>(n, z, 7 + r, d_out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms * 100.f;   // us per launch
}

int main(int argc, char** argv) {
  const double z = argc > 1 ? atof(argv[1]) : 0.25;
  const int64_t n = 1 << 20;
  double* d_out;
  hipMalloc(&d_out, sizeof(double) * n);
  printf("z = %g, n = %lld draws, one lane per draw, us per launch:\n", z, (long long)n);
  printf("  philox block + 2 uniforms   %8.1f\n", run<0>(n, z, d_out));
  printf("  mixture weights (right_mass) %7.1f\n", run<1>(n, z, d_out));
  printf("  series test                 %8.1f\n", run<2>(n, z, d_out));
  printf("  inverse-Gaussian loop       %8.1f\n", run<3>(n, z, d_out));
  printf("  one inverse-Gaussian attempt %7.1f\n", run<6>(n, z, d_out));
  printf("  exponential proposal        %8.1f\n", run<4>(n, z, d_out));
  printf("  whole draw (jacobi)         %8.1f\n", run<5>(n, z, d_out));
  return 0;
}
