// Polya-Gamma draws of the device chain, E elements per lane (chain.hip
// chain_pg_kernel; timed piece by piece in scripts/probes/pg_parts.hip).
//
// One lane per draw (rounds 1-4) leaves a wavefront in the truncated
// inverse-Gaussian rejection loop as long as its unluckiest lane: 4-8 attempts
// where a lane needs 1.4-2.1, 60-75 % of the kernel (profiles/r05_pg_rounds.txt).
// Here a lane owns E elements and walks the draw in three passes:
//   A  mixture weight, piece, the exponential-piece proposals   -- no loop
//   B  the inverse-Gaussian proposals of its elements that need one, one after
//      the other from a lane-private list: a wavefront now waits for the lane
//      with the largest SUM of attempts over E elements (relative spread
//      ~1/sqrt(E)), and no lane idles while it still has an element pending
//   C  the alternating-series test of the E proposals           -- no loop
// and the few proposals the series test rejects (< 1e-3) start over with the
// sequential sampler.  No barrier, no communication between lanes: the LDS
// below is lane-private storage indexed by element slot.
//
// Every piece of randomness has its own Philox sub-stream (`trial`) of the
// element: 0 the piece and the exponential proposal, 1 + k the k-th
// inverse-Gaussian attempt, 127 the series test, 128 a start-over -- a draw
// depends on (seed, stream, element) only, never on E, the grid or neighbours.
#pragma once

#include "philox.hpp"
#include "samplers.hpp"

namespace bbx {

constexpr int PG_TRIAL_SERIES = 127;
constexpr int PG_TRIAL_RESTART = 128;

// omega[i] ~ PG(n_trial[i], psi[i]) for the block's elements
// base + e * 256 + tid, e < E (n_trial: pointer to double or int32 shapes);
// returns the lane's sum of loglik(i, psi_i, n_trial_i).

// Elements per lane by problem size (scripts/probes/pg_parts.hip `whole`,
// profiles/r05_pg_queue.txt): the kernel is fastest with 250-500 blocks --
// one or two per CU -- 4 elements per lane from 50 000 draws on (n = 250k:
// 29-36 us against 50-57 with one), 8 from 800 000 (n = 1M: 61-84 us against
// 89-122; 97-153 for the one-lane kernel of rounds 1-4).
inline int polya_gamma_elems(int64_t n) {
  return n >= 800000 ? 8 : n >= 50000 ? 4 : 1;
}

template <int E, class Shape, class LogLik>
__device__ inline double polya_gamma_block(
    int64_t base, int64_t n, uint64_t seed, uint64_t stream, Shape n_trial,
    const double* __restrict__ psi, double* __restrict__ omega,
    double (*s_z)[256], double (*s_x)[256], LogLik loglik) {
  const int tid = threadIdx.x;
  double acc = 0.;
  unsigned fast = 0;   // bit e: element exists and is a single J*(1, z) draw
  unsigned need = 0;   // bit e: waits for an inverse-Gaussian proposal
  // ---- A
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int64_t i = base + (int64_t)e * 256 + tid;
    if (i >= n) continue;
    const double eta = psi[i];
    const double nt = (double)n_trial[i];
    acc += loglik(i, eta, nt);
    if (!(fabs(eta) <= 1.7e308)) {
      // a non-finite linear predictor (a failed draw upstream) must not spin in
      // a rejection loop whose every comparison is false: it propagates
      omega[i] = eta - eta;   // NaN
      continue;
    }
    if (nt != 1.) {
      // binomial outcomes: a sum of n_trial draws, the sequential sampler
      Philox g(seed, stream, (uint64_t)i);
      omega[i] = PolyaGamma::draw(g, (int)nt, eta);
      continue;
    }
    fast |= 1u << e;
    const double z = 0.5 * fabs(eta);
    s_z[e][tid] = z;
    const double rate = 0.5 * z * z + 0.125 * kPi * kPi;
    Philox g(seed, stream, (uint64_t)i, 0);
    if (g.uniform() < PolyaGamma::right_mass_direct(z, rate))
      s_x[e][tid] = PolyaGamma::trunc_exp(g, 1. / rate, PolyaGamma::kCut);
    else
      need |= 1u << e;
  }
  // ---- B
  unsigned att = 0;
  while (need) {
    const int e = __ffs(need) - 1;
    const int64_t i = base + (int64_t)e * 256 + tid;
    // attempts 0 .. 125 have a sub-stream each; a later attempt (probability
    // < 1e-19 per element) continues in sub-stream 126 further down its
    // counter, 64 blocks per attempt, so that it never replays a rejection
    Philox g(seed, stream, (uint64_t)i, 1u + (att < 125u ? att : 125u));
    if (att > 125u) g.ctr[0] += ((att - 125u) & 0x3FFFu) << 6;
    double x;
    if (PolyaGamma::trunc_inv_gauss_attempt(g, s_z[e][tid], PolyaGamma::kCut, x)) {
      s_x[e][tid] = x;
      need &= need - 1;
      att = 0;
    } else {
      att += 1;
    }
  }
  // ---- C
  unsigned redo = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (!(fast & (1u << e))) continue;
    const int64_t i = base + (int64_t)e * 256 + tid;
    Philox g(seed, stream, (uint64_t)i, PG_TRIAL_SERIES);
    const double x = s_x[e][tid];
    if (PolyaGamma::series_accept_direct(g, x))
      omega[i] = 0.25 * x;
    else
      redo |= 1u << e;
  }
  while (redo) {   // (rare: the proposal's envelope is tight)
    const int e = __ffs(redo) - 1;
    redo &= redo - 1;
    const int64_t i = base + (int64_t)e * 256 + tid;
    Philox g(seed, stream, (uint64_t)i, PG_TRIAL_RESTART);
    omega[i] = 0.25 * PolyaGamma::jacobi(g, s_z[e][tid]);
  }
  return acc;
}

}  // namespace bbx
