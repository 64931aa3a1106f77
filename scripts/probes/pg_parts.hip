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
#include <string.h>

#include "philox.hpp"
#include "samplers.hpp"
#include "pg_queue.hpp"

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

// ---- whole kernels on a vector of psi: one lane per draw against E draws per
// lane (csrc/pg_queue.hpp), with the sample mean / variance of omega
__global__ __launch_bounds__(256) void whole_lane_kernel(
    int64_t n, uint64_t seed, const double* __restrict__ psi,
    double* __restrict__ omega) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256) {
    Philox g(seed, 3, (uint64_t)i);
    omega[i] = PolyaGamma::draw(g, 1, psi[i]);
  }
}

template <int E>
__global__ __launch_bounds__(256) void whole_queue_kernel(
    int64_t n, uint64_t seed, const double* __restrict__ n_trial,
    const double* __restrict__ psi, double* __restrict__ omega,
    double* __restrict__ sink) {
  __shared__ double s_z[E][256], s_x[E][256];
  double acc = 0.;
  for (int64_t base = (int64_t)blockIdx.x * 256 * E; base < n;
       base += (int64_t)gridDim.x * 256 * E)
    acc += polya_gamma_block<E>(base, n, seed, 3, n_trial, psi, omega, s_z, s_x,
                                [](int64_t, double eta, double nt) {
                                  return nt * eta;
                                });
  if (acc == 1.2345) sink[0] = acc;
}

static void moments(const double* d, int64_t n, double* mean, double* var) {
  static double* h = nullptr;
  if (!h) h = (double*)malloc(sizeof(double) * n);
  hipMemcpy(h, d, sizeof(double) * n, hipMemcpyDeviceToHost);
  double s = 0., s2 = 0.;
  for (int64_t i = 0; i < n; ++i) s += h[i];
  s /= (double)n;
  for (int64_t i = 0; i < n; ++i) s2 += (h[i] - s) * (h[i] - s);
  *mean = s;
  *var = s2 / (double)n;
}

template <class F>
static float time_us(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch(7);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 10; ++r) launch(8 + r);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms * 100.f;
}

static void whole_kernels(int64_t n, double psi_sd) {
  // psi ~ N(0, psi_sd^2) (psi_sd < 0: every |psi| = -psi_sd)
  double* h = (double*)malloc(sizeof(double) * n);
  uint64_t st = 12345;
  for (int64_t i = 0; i < n; ++i) {
    double a = 0.;
    for (int k = 0; k < 12; ++k) {
      st = st * 6364136223846793005ull + 1442695040888963407ull;
      a += (double)(st >> 11) / 9007199254740992.0;
    }
    h[i] = psi_sd < 0. ? -psi_sd : (a - 6.) * psi_sd;
  }
  double *d_psi, *d_nt, *d_om, *d_sink;
  hipMalloc(&d_psi, sizeof(double) * n);
  hipMalloc(&d_nt, sizeof(double) * n);
  hipMalloc(&d_om, sizeof(double) * n);
  hipMalloc(&d_sink, 64);
  hipMemcpy(d_psi, h, sizeof(double) * n, hipMemcpyHostToDevice);
  for (int64_t i = 0; i < n; ++i) h[i] = 1.;
  hipMemcpy(d_nt, h, sizeof(double) * n, hipMemcpyHostToDevice);
  free(h);
  double m, v;
  const int grid_lane = (int)((n + 255) / 256);
  float us = time_us([&](int s) {
    whole_lane_kernel<<<grid_lane < 4096 ? grid_lane : 4096, 256>>>(n, s, d_psi, d_om);
  });
  moments(d_om, n, &m, &v);
  printf("  psi sd %5.2f  one lane per draw (grid 4096)  %7.1f us   mean %.6f var %.6f\n",
         psi_sd, us, m, v);
#define QUEUE(EE)                                                              \
  {                                                                            \
    const int grid = (int)((n + 256 * EE - 1) / (256 * EE));                   \
    us = time_us([&](int s) {                                                  \
      whole_queue_kernel<EE><<<grid, 256>>>(n, s, d_nt, d_psi, d_om, d_sink);  \
    });                                                                        \
    moments(d_om, n, &m, &v);                                                  \
    printf("  psi sd %5.2f  %2d draws per lane (grid %5d)   %7.1f us   mean %.6f var %.6f\n", \
           psi_sd, EE, grid, us, m, v);                                        \
  }
  QUEUE(1) QUEUE(2) QUEUE(4) QUEUE(8) QUEUE(16)
#undef QUEUE
  hipFree(d_psi);
  hipFree(d_nt);
  hipFree(d_om);
  hipFree(d_sink);
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "whole")) {
    const int64_t n = argc > 2 ? atoll(argv[2]) : 1000000;
    printf("n = %lld Polya-Gamma(1, psi) draws, whole kernels:\n", (long long)n);
    for (double sd : {-0.1, -0.5, -2., -5., 0.5, 1.5, 3.}) whole_kernels(n, sd);
    return 0;
  }
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
