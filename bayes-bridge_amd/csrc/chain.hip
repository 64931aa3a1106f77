// Device-resident Gibbs chain: everything BayesBridge.gibbs does per iteration
// (bayesbridge.py:210-240) around the CG draw, as kernels on the design's
// stream.  The scalar samplers are the shared templates of samplers.hpp driven
// by Philox counters.
#include <cmath>
#include <new>
#include <vector>

#include "chain.hpp"
#include "philox.hpp"
#include "samplers.hpp"
#include "pg_queue.hpp"

namespace bbx {

__device__ inline double wsum(double x) { return wave_allsum(x); }

__device__ inline double block_total_256(double x) {
  __shared__ double s_w[256 / WAVE];
  x = wsum(x);
  if ((threadIdx.x & (WAVE - 1)) == 0) s_w[threadIdx.x / WAVE] = x;
  __syncthreads();
  double r = 0.;
  if (threadIdx.x == 0)
    for (int k = 0; k < 256 / WAVE; ++k) r += s_w[k];
  __syncthreads();
  return r;  // thread 0 only
}

__device__ inline double shrunk_scale(double gscale, double lscale,
                                      double slab) {
  // reg_coef_sampler.py:194-201
  const double raw = gscale * lscale;
  const double ratio = raw / slab;
  return raw / sqrt(1. + ratio * ratio);
}

// phi, CG warm start, preconditioner sd, z  (reg_coef_sampler.py:74-89;
// reg_coef_posterior_summarizer.py:25-29,105-124).
__global__ __launch_bounds__(256) void chain_prior_kernel(
    int64_t P, int nu, int model, double slab, long long n_avg,
    const ChainScalars* __restrict__ sc, const double* __restrict__ lscale,
    const double* __restrict__ sd_unshrunk, const double* __restrict__ mean,
    const double* __restrict__ square, const double* __restrict__ zbase,
    double* __restrict__ phi, double* __restrict__ x0, double* __restrict__ sd,
    double* __restrict__ z, int store_idx, double* __restrict__ samp_gs,
    double* __restrict__ samp_lp) {
  const double g = sc->gscale;
  const double zs = (model == BBX_MODEL_LINEAR) ? sc->obs_prec : 1.;
  // global scale and log posterior of the PREVIOUS iteration, if it was kept
  // (chain_save_sample defers them to here: the scalars are untouched between
  // the end of an iteration and this kernel, and a launch of its own for two
  // stores sat on the serial stretch in front of every solve)
  if (store_idx >= 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    samp_gs[store_idx] = g;
    samp_lp[store_idx] = sc->logp();
  }
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < P;
       j += (int64_t)gridDim.x * 256) {
    double prior_sd, guess = mean[j];
    if (j < nu) {
      prior_sd = sd_unshrunk[j];
    } else {
      prior_sd = shrunk_scale(g, lscale[j - nu], slab);
      guess *= prior_sd;
    }
    phi[j] = 1. / prior_sd;
    x0[j] = guess;
    double est = 1.;
    if (n_avg > 1) {
      const double k = (double)n_avg;
      const double m = mean[j];
      const double var = k / (k - 1.) * (square[j] - m * m);
      const double w = (k - 1.) / (k - 1. + 5.);
      est = sqrt(w * var + (1. - w) * 1.);
    }
    sd[j] = est;
    z[j] = zs * zbase[j];
  }
}

// Running mean / second moment of the scaled coefficients
// (reg_coef_posterior_summarizer.py:11-19,93-103).
// The three small kernels that lead the tau / lambda branch are a serial prefix
// of the longest path after a draw (a single wave's gamma draw takes 51 us under
// the pass and the Polya-Gamma kernel, 10 us alone): they ask the wave scheduler
// for priority.  (BBX_BRANCH_PRIO=0: A/B.)
#define BBX_BRANCH_PRIO() do { if (prio) __builtin_amdgcn_s_setprio(3); } while (0)
static int lscale_prio() {
  static const int on = getenv("BBX_LSCALE_PRIO") ? atoi(getenv("BBX_LSCALE_PRIO")) : 1;
  return on;
}
static int branch_prio() {
  static const int on = !(getenv("BBX_BRANCH_PRIO") && atoi(getenv("BBX_BRANCH_PRIO")) == 0);
  return on;
}
__global__ __launch_bounds__(256) void chain_summary_kernel(
    int64_t P, int nu, double slab, long long n_avg,
    const ChainScalars* __restrict__ sc, const double* __restrict__ lscale,
    const double* __restrict__ coef, double* __restrict__ mean,
    double* __restrict__ square, int prio) {
  BBX_BRANCH_PRIO();
  const double g = sc->gscale;
  const double w = 1. / (1. + (double)n_avg);
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < P;
       j += (int64_t)gridDim.x * 256) {
    double theta = coef[j];
    if (j >= nu) theta /= shrunk_scale(g, lscale[j - nu], slab);
    mean[j] = w * theta + (1. - w) * mean[j];
    square[j] = w * theta * theta + (1. - w) * square[j];
  }
}

__device__ inline double log1pexp(double x) {
  // np.logaddexp(0, x)  (logistic_model.py:52-55)
  return x > 0. ? x + log1p(exp(-x)) : log1p(exp(x));
}

// Omega_i ~ PG(n_trial_i, psi_i) and the log-likelihood partials
// (bayesbridge.py:405-408; logistic_model.py:49-55).  E elements per lane:
// pg_queue.hpp.
template <int E>
__global__ __launch_bounds__(256) void chain_pg_kernel(
    int64_t n, uint64_t seed, uint64_t stream,
    const double* __restrict__ n_success, const double* __restrict__ n_trial,
    const double* __restrict__ psi, double* __restrict__ omega,
    double* __restrict__ ll_part) {
  __shared__ double s_z[E][256], s_x[E][256];
  double acc = 0.;
  for (int64_t base = (int64_t)blockIdx.x * (256 * E); base < n;
       base += (int64_t)gridDim.x * (256 * E))
    acc += polya_gamma_block<E>(
        base, n, seed, stream, n_trial, psi, omega, s_z, s_x,
        [&](int64_t i, double eta, double nt) {
          return n_success[i] * eta - nt * log1pexp(eta);
        });
  const double tot = block_total_256(acc);
  if (threadIdx.x == 0) ll_part[blockIdx.x] = tot;
}

// (one lane per draw with the sequential sampler: the kernel of rounds 1-4,
// kept behind BBX_PG_ELEMS=0 for A/B runs)
__global__ __launch_bounds__(256) void chain_pg_lane_kernel(
    int64_t n, uint64_t seed, uint64_t stream,
    const double* __restrict__ n_success, const double* __restrict__ n_trial,
    const double* __restrict__ psi, double* __restrict__ omega,
    double* __restrict__ ll_part) {
  double acc = 0.;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256) {
    const double eta = psi[i];
    const double nt = n_trial[i];
    Philox g(seed, stream, (uint64_t)i);
    omega[i] = fabs(eta) <= 1.7e308 ? PolyaGamma::draw(g, (int)nt, eta) : eta - eta;
    acc += n_success[i] * eta - nt * log1pexp(eta);
  }
  const double tot = block_total_256(acc);
  if (threadIdx.x == 0) ll_part[blockIdx.x] = tot;
}

// Polya-Gamma mean at the current linear predictor (logistic_model.py:80-87).
__global__ __launch_bounds__(256) void chain_pg_mean_kernel(
    int64_t n, const double* __restrict__ n_trial,
    const double* __restrict__ psi, double* __restrict__ omega) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256) {
    const double t = psi[i];
    double m = n_trial[i] / 2.;
    if (fabs(t) > 1e-5) m *= 1. / t * (exp(t) - 1.) / (exp(t) + 1.);
    omega[i] = m;
  }
}

// Residual sum of squares partials (linear model; bayesbridge.py:400-403).
__global__ __launch_bounds__(256) void chain_rss_kernel(
    int64_t n, const double* __restrict__ y, const double* __restrict__ psi,
    double* __restrict__ part) {
  double acc = 0.;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256) {
    const double r = y[i] - psi[i];
    acc += r * r;
  }
  const double tot = block_total_256(acc);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__device__ inline double sum_row_parts(const double* part, int count) {
  // one block, fixed order
  double acc = 0.;
  for (int k = threadIdx.x; k < count; k += 256) acc += part[k];
  return block_total_256(acc);
}

// Finishes the observation-level reductions (single block).
//   logit : loglik = sum of partials
//   linear: obs_prec ~ Gamma(n/2, 1) / (rss/2); loglik = n log(w)/2 - w rss/2
//           (bayesbridge.py:400-404; linear_model.py:13-17); `init` => 1/mean
__global__ __launch_bounds__(256) void chain_obs_finish_kernel(
    int model, int init, int64_t n, uint64_t seed, uint64_t stream,
    const double* __restrict__ part, int count, ChainScalars* __restrict__ sc) {
  const double tot = sum_row_parts(part, count);
  if (threadIdx.x != 0) return;
  if (model == BBX_MODEL_LOGIT) {
    sc->loglik = tot;
  } else {
    double w;
    if (init) {
      w = 1. / (tot / (double)n);
    } else {
      Philox g(seed, stream, 0);
      const double obs_var = (tot / 2.) / gamma_draw(g, (double)n / 2.);
      w = 1. / obs_var;
    }
    sc->obs_prec = w;
    sc->loglik = (double)n * log(w) / 2. - w * tot / 2.;
  }
}

__global__ __launch_bounds__(256) void chain_fill_obs_prec_kernel(
    int64_t n, const ChainScalars* __restrict__ sc, double* __restrict__ omega) {
  const double w = sc->obs_prec;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256)
    omega[i] = w;
}

// Partials of sum |beta_j|^alpha (shrunk), sum (beta_j/slab)^2 (all) and
// sum (beta_j/sd_j)^2 (unshrunk).  Grid = NPART.
__global__ __launch_bounds__(256) void chain_coef_sums_kernel(
    int64_t P, int nu, double alpha, double slab,
    const double* __restrict__ coef, const double* __restrict__ sd_unshrunk,
    double* __restrict__ part_pow, double* __restrict__ part_slab,
    double* __restrict__ part_fixed, int prio) {
  BBX_BRANCH_PRIO();
  double a = 0., b = 0., c = 0.;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < P;
       j += (int64_t)gridDim.x * 256) {
    const double x = coef[j];
    const double xs = x / slab;
    b += xs * xs;
    if (j >= nu) {
      a += pow(fabs(x), alpha);
    } else {
      const double xf = x / sd_unshrunk[j];
      c += xf * xf;
    }
  }
  const double ta = block_total_256(a);
  const double tb = block_total_256(b);
  const double tc = block_total_256(c);
  if (threadIdx.x == 0) {
    part_pow[blockIdx.x] = ta;
    part_slab[blockIdx.x] = tb;
    part_fixed[blockIdx.x] = tc;
  }
}

// tau | beta: conjugate Gamma update of tau^-alpha (bayesbridge.py:412-448)
// and, with the new tau, the log posterior (bayesbridge.py:480-511).
__global__ __launch_bounds__(256) void chain_gscale_kernel(
    int64_t n_shrunk, int nu, double alpha, double shape0, double rate0,
    double lower_bd, int update_mode, uint64_t seed, uint64_t stream,
    const double* __restrict__ part_pow, const double* __restrict__ part_slab,
    const double* __restrict__ part_fixed,
    const double* __restrict__ sd_unshrunk, ChainScalars* __restrict__ sc,
    int prio) {
  BBX_BRANCH_PRIO();
  const double pow_sum = sum_row_parts(part_pow, NPART);
  const double slab_sum = sum_row_parts(part_slab, NPART);
  const double fixed_sum = sum_row_parts(part_fixed, NPART);
  if (threadIdx.x != 0) return;
  double g = sc->gscale;
  if (n_shrunk > 0) {
    if (update_mode == BBX_GSCALE_OPTIMIZE) {
      // Monte-Carlo EM step (bayesbridge.py:450-456)
      const double ph = (double)n_shrunk / alpha / pow_sum;
      g = pow(ph, -(1. / alpha));
    } else if (update_mode == BBX_GSCALE_SAMPLE) {
      if (pow_sum == 0.) {
        g = 0.;  // bayesbridge.py:430-431
      } else {
        const double shape = shape0 + (double)n_shrunk / alpha;
        const double rate = rate0 + pow_sum;
        Philox rng(seed, stream, 0);
        const double ph = gamma_draw(rng, shape) / rate;
        g = 1. / pow(ph, 1. / alpha);
      }
    }
    // method None keeps tau and skips the bound (bayesbridge.py:441)
    if (update_mode != BBX_GSCALE_FIXED && g < lower_bd) {
      g = lower_bd;
      sc->n_gscale_clamped += 1;
    }
  } else {
    g = 1.;  // bayesbridge.py:417-418 placeholder
  }
  sc->gscale = g;
  sc->abs_pow_sum = pow_sum;
  // log posterior with the NEW tau (bayesbridge.py:232-234,480-511), without
  // the log-likelihood, which the Omega branch writes (ChainScalars::logp)
  double lp = -.5 * slab_sum;
  double prior = 0.;
  if (n_shrunk > 0)
    prior += -(double)n_shrunk * log(g) - pow_sum / pow(g, alpha);
  prior += -.5 * fixed_sum;
  for (int j = 0; j < nu; ++j)
    if (sd_unshrunk[j] < INFINITY) prior -= log(sd_unshrunk[j]);
  prior += (shape0 - 1.) * log(g) - rate0 * g;
  sc->logprior = lp + prior;
}

// lambda_j | tau, beta_j (bayesbridge.py:458-478).
//
// lambda_j^-2 / 2 is an exponentially tilted stable variate sampled by
// rejection (tilted_stable.pyx:99-104: plain rejection when tilt^a < 2,
// Devroye's double rejection otherwise).  One proposal is a few thousand
// dependent f64 instructions, and with one lane per coefficient a wavefront
// runs as long as its unluckiest lane (~10 proposals; 0.39 ms at p = 5e4).
// Here a 256-thread block owns 256 coefficients and works in rounds: the
// still-pending coefficients are compacted in LDS and the lanes they leave idle
// evaluate further candidate proposals of the same coefficients, each on its
// own Philox sub-stream (seed, iteration, j, candidate number).  A coefficient
// takes its lowest-numbered accepted candidate, which is exactly what
// proposing candidates 0, 1, 2, ... one after the other would return, so the
// draw depends on neither the round structure nor the scheduling.
constexpr int TS_BLOCK = 256;
constexpr int TS_MAX_COPIES = 16;   // candidates per coefficient and round
constexpr unsigned TS_MAX_TRIAL = 4095u;  // Philox sub-stream budget

template <class Store>
__device__ inline void tilted_stable_block(int64_t base, int64_t count,
                                           double a, uint64_t seed,
                                           uint64_t stream,
                                           const double* s_tilt,
                                           double cost_threshold, Store store) {
  // s_tilt[i]: tilt of local item i (< count <= TS_BLOCK), already in LDS
  //
  // The two regimes of the sampler (plain rejection when tilt^a < 2, double
  // rejection otherwise) are different code of very different length; lanes of
  // both kinds in one wavefront execute both, one after the other.  The list
  // of pending items is therefore kept PARTITIONED -- double-rejection items
  // first, plain-rejection items after them -- so that all but one wavefront
  // of a round run a single regime.  Which lane evaluates which (item, trial)
  // pair has no influence on the draw (each pair has its own Philox
  // sub-stream and the lowest accepted trial wins).
  __shared__ int s_pending[TS_BLOCK];
  __shared__ int s_next_dr[TS_BLOCK];
  __shared__ int s_next_dc[TS_BLOCK];
  __shared__ unsigned s_tried[TS_BLOCK];   // candidates already evaluated
  __shared__ int s_winner[TS_BLOCK];       // lowest accepted copy this round
  __shared__ int s_n_dr, s_n_dc;
  const int tid = threadIdx.x;
  const double odds = (1. - a) / a;
  if (tid == 0) {
    s_n_dr = 0;
    s_n_dc = 0;
  }
  __syncthreads();
  if (tid < count) {
    s_tried[tid] = 0u;
    const bool cheap = pos_pow(s_tilt[tid], a) < fabs(cost_threshold);
    if (cheap)
      s_next_dc[atomicAdd(&s_n_dc, 1)] = tid;
    else
      s_next_dr[atomicAdd(&s_n_dr, 1)] = tid;
  }
  __syncthreads();
  for (;;) {
    // pending = [double-rejection items | plain-rejection items]
    const int n_dr = s_n_dr, m = n_dr + s_n_dc;
    if (m == 0) break;
    if (tid < m)
      s_pending[tid] = tid < n_dr ? s_next_dr[tid] : s_next_dc[tid - n_dr];
    __syncthreads();
    int copies = TS_BLOCK / m;
    if (copies > TS_MAX_COPIES) copies = TS_MAX_COPIES;
    // lane -> (item slot q, candidate c): consecutive lanes take consecutive
    // slots, so a wavefront covers one stretch of the partitioned list
    const int q = tid % m, c = tid / m;
    const bool active = c < copies;
    const int item = s_pending[q];
    if (tid < m) s_winner[s_pending[tid]] = 0x7fffffff;
    if (tid == 0) {
      s_n_dr = 0;
      s_n_dc = 0;
    }
    __syncthreads();
    bool ok = false, cheap = false;
    double val = 0.;
    if (active) {
      const double tilt = s_tilt[item];
      const double tilt_pow = pos_pow(tilt, a);
      cheap = tilt_pow < fabs(cost_threshold);
      unsigned trial = s_tried[item] + (unsigned)c;
      if (trial > TS_MAX_TRIAL) trial = TS_MAX_TRIAL;
      Philox rng(seed, stream, (uint64_t)(base + item), trial);
      if (cheap) {
        // tilt^a < 2 => a single part, c = 1 (tilted_stable.pyx:138-140)
        ok = TiltedStable::dc_trial(rng, a, tilt, 1., val);
      } else {
        double x;
        ok = (cost_threshold < 0.) ? TiltedStable::dr_trial(rng, a, tilt_pow, x)
                                   : TiltedStable::dr_trial_flat(rng, a, tilt_pow, x);
        val = pos_pow(x, -odds);
      }
      if (trial >= TS_MAX_TRIAL) ok = true;  // budget exhausted: keep it
      if (ok) atomicMin(&s_winner[item], c);
    }
    __syncthreads();
    if (active && ok && s_winner[item] == c) store(item, val);
    if (tid < m) {
      const int it = s_pending[tid];
      if (s_winner[it] == 0x7fffffff) {
        s_tried[it] += (unsigned)copies;
        if (tid < n_dr)
          s_next_dr[atomicAdd(&s_n_dr, 1)] = it;
        else
          s_next_dc[atomicAdd(&s_n_dc, 1)] = it;
      }
    }
    __syncthreads();
  }
}

// `items` (<= TS_BLOCK) coefficients per pass of a 256-thread block: with fewer
// items than threads every coefficient starts with TS_BLOCK/items candidate
// proposals at once.  The accepted value is the one of the LOWEST accepted trial
// index, so the result does not depend on `items` (or on the grid).
__global__ __launch_bounds__(TS_BLOCK) void chain_lscale_kernel(
    int64_t n_shrunk, int nu, double alpha, uint64_t seed, uint64_t stream,
    ChainScalars* __restrict__ sc, const double* __restrict__ coef,
    double* __restrict__ lscale, int items, double cost_threshold, int prio) {
  __shared__ double s_tilt[TS_BLOCK];
  // (beside the Polya-Gamma kernel; three alternating pairs: config 2 36.6-36.9
  // against 37.1-37.2 us per CG iteration, config 3 level; BBX_LSCALE_PRIO=0: A/B)
  BBX_BRANCH_PRIO();
  const double g = sc->gscale;
  for (int64_t base = (int64_t)blockIdx.x * items; base < n_shrunk;
       base += (int64_t)gridDim.x * items) {
    const int64_t count =
        (n_shrunk - base < items) ? (n_shrunk - base) : items;
    if (alpha == 2.) {
      if (threadIdx.x < count) lscale[base + threadIdx.x] = .5;  // :460-461
      continue;
    }
    if (threadIdx.x < count) {
      const double r = coef[base + threadIdx.x + nu] / g;
      s_tilt[threadIdx.x] = r * r;
    }
    __syncthreads();
    tilted_stable_block(base, count, alpha / 2., seed, stream, s_tilt,
                        cost_threshold, [&](int item, double ts) {
                          double l = sqrt(.5 / ts);
                          if (l == 0.) {
                            l = 10e-16;  // bayesbridge.py:470-472
                          } else if (isinf(l)) {
                            l = 2.0 / g;  // bayesbridge.py:473-476
                          }
                          lscale[base + item] = l;
                        });
    __syncthreads();
  }
}

__global__ void chain_store_scalars_kernel(const ChainScalars* __restrict__ sc,
                                           int idx,
                                           double* __restrict__ gs,
                                           double* __restrict__ lp) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    gs[idx] = sc->gscale;
    lp[idx] = sc->logp();
  }
}

// kappa = n_success - n_trial/2 (bayesbridge.py:380 with Omega cancelled).
__global__ __launch_bounds__(256) void chain_kappa_kernel(
    int64_t n, const double* __restrict__ ns, const double* __restrict__ nt,
    double* __restrict__ kappa) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256)
    kappa[i] = ns[i] - nt[i] / 2.;
}

// -------------------------------------------- stand-alone sampler kernels

// (the chain's sampler, pg_queue.hpp: what the distribution tests draw from)
template <int E>
__global__ __launch_bounds__(256) void dev_pg_kernel(
    int64_t n, uint64_t seed, const int32_t* __restrict__ shape,
    const double* __restrict__ tilt, double* __restrict__ out) {
  __shared__ double s_z[E][256], s_x[E][256];
  for (int64_t base = (int64_t)blockIdx.x * (256 * E); base < n;
       base += (int64_t)gridDim.x * (256 * E))
    polya_gamma_block<E>(base, n, seed, STREAM_PG, shape, tilt, out, s_z, s_x,
                         [](int64_t, double, double) { return 0.; });
}

__global__ __launch_bounds__(TS_BLOCK) void dev_ts_kernel(
    int64_t n, uint64_t seed, double a, const double* __restrict__ tilt,
    double* __restrict__ out, double cost_threshold) {
  __shared__ double s_tilt[TS_BLOCK];
  for (int64_t base = (int64_t)blockIdx.x * TS_BLOCK; base < n;
       base += (int64_t)gridDim.x * TS_BLOCK) {
    const int64_t count = (n - base < TS_BLOCK) ? (n - base) : TS_BLOCK;
    if (threadIdx.x < count) s_tilt[threadIdx.x] = tilt[base + threadIdx.x];
    __syncthreads();
    tilted_stable_block(base, count, a, seed, STREAM_LSCALE, s_tilt,
                        cost_threshold,
                        [&](int item, double ts) { out[base + item] = ts; });
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void dev_gamma_kernel(
    int64_t n, uint64_t seed, double shape, double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256) {
    Philox g(seed, STREAM_GSCALE, (uint64_t)i);
    out[i] = gamma_draw(g, shape);
  }
}

// ------------------------------------------------------------ host helpers

// Where the tilted-stable sampler switches from plain rejection (expected
// e^{tilt^a} proposals of ~1 us-equivalents each, evaluated many at a time) to
// Devroye's double rejection (a bounded number of MUCH longer proposals with
// inner rejection loops).  The reference switches at tilt^a = 2
// (tilted_stable.pyx:53,99-104), a figure tuned for one scalar core; both
// methods are exact samplers of the same law, so the switch point is a pure
// cost choice.  Measured on MI355X, 50 000 draws at one tilt (ts_regimes.py):
// plain rejection 33 / 45 / 71 us at tilt^a = .1 / 1 / 1.9, double rejection
// 290-380 us at 2.1-4 and 130 / 110 us at 16 / 100.
// (a negative value selects the in-place inner loop of dr_trial)
static double ts_cost_threshold() { return TiltedStable::kCostThreshold; }

static inline uint64_t iter_stream(uint64_t stream, int64_t iter) {
  return stream | ((uint64_t)iter << 8);
}

static int grid_for(int64_t len, int cap) {
  int64_t nb = (len + 255) / 256;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// psi = X~ coef
static int chain_linear_predictor(bbx_chain* c) {
  bbx_design* h = c->h;
  // (h->skip_flag: null outside a CG solve; inside one -- the speculative form
  // below -- the flag both kernels return on)
  BBX_TRY(launch_prep_v(h, c->coef.as<double>(), nullptr, nullptr,
                        part_slot(h, PS_C), h->skip_flag));
  return launch_dot(h, c->coef.as<double>(), nullptr, c->psi.as<double>(),
                    nullptr);
}

// The pass for X~ beta enqueued by the CG loop itself, right behind a look at
// the stop flag (bbx_design::tail_hook): when the host wakes up from that look
// the pass is already running, and the kernels of both branches are launched
// under it instead of after a 25-35 us gap.  If the rule had not fired the two
// kernels returned at entry.  Layouts whose X~ v is a single kernel that honours
// the skip flag: the tiled format, one column group, not split by value.
static int chain_tail_hook(void* ctx) {
  return chain_linear_predictor(static_cast<bbx_chain*>(ctx));
}

static bool chain_tail_applies(const bbx_chain* c) {
  static const bool on = !(getenv("BBX_CHAIN_TAIL") && atoi(getenv("BBX_CHAIN_TAIL")) == 0);
  const bbx_design* h = c->h;
  if (!on || !h->sparse || h->format != BBX_FORMAT_TILED || h->hybrid) return false;
  int G = 0;
  return tiled_describe(h, 0, nullptr, nullptr, nullptr, &G, nullptr, nullptr) ==
             BBX_OK && G == 1;
}

static int chain_check(const bbx_chain* c) {
  if (!c || !c->h) return fail(BBX_ERR_INVALID, "chain handle is NULL");
  return BBX_OK;
}

static double power_exp_ave_magnitude(double exponent) {
  // prior.py:163-167
  return std::tgamma(2. / exponent) / std::tgamma(1. / exponent);
}

int chain_pre_draw(bbx_chain* c) {
  bbx_design* h = c->h;
  hipStream_t s = h->stream;
  const int64_t P = h->P, n = h->n;
  ChainScalars* sc = c->scalars.as<ChainScalars>();
  // --- beta | Omega, tau, lambda  (bayesbridge.py:372-395)
  BBX_LAUNCH(chain_prior_kernel, dim3(NPART), dim3(256), 0, s, P,
                     c->n_unshrunk, c->model, c->slab,
                     (long long)c->n_averaged, sc, c->lscale.as<double>(),
                     c->sd_unshrunk.as<double>(), c->mean.as<double>(),
                     c->square.as<double>(), c->zbase.as<double>(),
                     c->phi.as<double>(), c->x0.as<double>(),
                     c->sd.as<double>(), c->z.as<double>(), c->pending_store,
                     c->samp_gscale.as<double>(), c->samp_logp.as<double>());
  c->pending_store = -1;
  if (c->model == BBX_MODEL_LINEAR)
    BBX_LAUNCH(chain_fill_obs_prec_kernel, dim3(grid_for(n, ROW_GRID)),
                       dim3(256), 0, s, n, sc, c->obs_prec.as<double>());
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int chain_post_draw(bbx_chain* c, bool have_psi, int phases,
                    bbx_chain* branch_of) {
  bbx_design* h = c->h;
  hipStream_t s = h->stream;
  const int64_t P = h->P, n = h->n;
  const int nu = c->n_unshrunk;
  const int64_t n_shrunk = P - nu;
  ChainScalars* sc = c->scalars.as<ChainScalars>();

  // The two updates that follow read beta and nothing of each other: the
  // Omega branch (X~ beta: one pass over the matrix, then n Polya-Gamma draws)
  // stays on the design's stream, the tau / lambda branch (three small
  // kernels, the last one ~0.1 ms of latency-bound rejection sampling on p
  // coefficients) runs beside it on a second stream.  Each branch writes its
  // own half of ChainScalars; Philox streams are keyed by element, so the
  // draws do not depend on the interleaving.  beta is final in the ORDER OF
  // THE DESIGN'S STREAM only: cg_sample_device returns with its finish kernel
  // enqueued and `ev_poll` recorded behind it (coef_in_flight), so the second
  // stream waits on that event before its first kernel (the summary kernel,
  // which reads beta and the OLD tau and lambda).  The design's stream waits
  // for the branch at the end.  Measured: config 3 +1.6 %, config 2 +2.5 %, config 4
  // +-0; tiny problems keep one stream, and so should processes that SHARE a
  // GPU (two ranks on one device ran 3x slower with a second queue each:
  // chains.py sets BBX_CHAIN_FORK=0 then).  BBX_CHAIN_FORK=0 / 1 forces one /
  // two streams.
  static const int fork_env =
      getenv("BBX_CHAIN_FORK") ? atoi(getenv("BBX_CHAIN_FORK")) : -1;
  const bool fork = fork_env >= 0 ? fork_env == 1
                                  : (n >= 50000 && n_shrunk >= 2048);
  bbx_chain* owner = branch_of ? branch_of : c;   // whose second stream carries the branch
  if (fork && owner->stream2 == nullptr) {
    // created on first use: chains that never fork keep a single queue
    BBX_HIP(hipStreamCreateWithFlags(&owner->stream2, hipStreamNonBlocking));
  }
  if (fork && c->ev_join == nullptr)
    BBX_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  hipStream_t s_b = !fork ? s : owner->stream2;
  if (h->coef_in_flight) {
    if (fork && (phases & POST_BRANCH))
      BBX_HIP(hipStreamWaitEvent(s_b, h->ev_poll, 0));
    h->coef_in_flight = false;
  }

  // --- Omega | beta  (bayesbridge.py:397-410).  Launch order = what has to
  // start first: the pass over the matrix, then the whole second branch (it
  // starts under that pass: the lambda kernel and the Polya-Gamma kernel are
  // both ALU-bound and slow each other down, the pass is bandwidth-bound),
  // then the n Polya-Gamma draws.
  if (!have_psi && (phases & POST_BRANCH)) BBX_TRY(chain_linear_predictor(c));

  if (phases & POST_BRANCH) {
  // --- running summaries of beta (with the tau and lambda it was drawn
  // under), then tau | beta, then lambda | tau, beta, then log posterior
  BBX_LAUNCH(chain_summary_kernel, dim3(NPART), dim3(256), 0, s_b, P,
                     nu, c->slab, (long long)c->n_averaged, sc,
                     c->lscale.as<double>(), c->coef.as<double>(),
                     c->mean.as<double>(), c->square.as<double>(),
                     fork ? branch_prio() : 0);
  c->n_averaged += 1;
  // the chain's own partial slots: the branches of a batch's chains run side by side
  double* pp = c->misc_part.as<double>();
  BBX_LAUNCH(chain_coef_sums_kernel, dim3(NPART), dim3(256), 0, s_b, P,
                     nu, c->bridge_exp, c->slab, c->coef.as<double>(),
                     c->sd_unshrunk.as<double>(), pp, pp + NPART,
                     pp + 2 * NPART, fork ? branch_prio() : 0);
  const double lower_bd = .001 / power_exp_ave_magnitude(c->bridge_exp);
  BBX_LAUNCH(chain_gscale_kernel, dim3(1), dim3(256), 0, s_b, n_shrunk,
                     nu, c->bridge_exp, c->shape0, c->rate0, lower_bd,
                     c->gscale_update, c->seed,
                     iter_stream(STREAM_GSCALE, c->iter), pp, pp + NPART,
                     pp + 2 * NPART, c->sd_unshrunk.as<double>(), sc,
                     fork ? branch_prio() : 0);
  if (n_shrunk > 0) {
    // Measured at p = 50k (ms per Gibbs iteration, items per block): 256:
    // 5.69, 128: 5.63, 64: 5.72, 32: 5.83, 16: 6.11 -- the speculative copies
    // of small blocks cost more arithmetic than the extra blocks hide latency.
    static const int items_env =
        getenv("BBX_TS_ITEMS") ? atoi(getenv("BBX_TS_ITEMS")) : 0;
    const int items = (items_env >= 8 && items_env <= TS_BLOCK) ? items_env
                      : (n_shrunk / TS_BLOCK < 1024) ? 128 : TS_BLOCK;
    int64_t nb = (n_shrunk + items - 1) / items;
    if (nb > 8192) nb = 8192;
    BBX_LAUNCH(chain_lscale_kernel, dim3((unsigned)nb), dim3(TS_BLOCK),
                       0, s_b, n_shrunk, nu, c->bridge_exp, c->seed,
                       iter_stream(STREAM_LSCALE, c->iter), sc,
                       c->coef.as<double>(), c->lscale.as<double>(), items,
                       ts_cost_threshold(), fork ? lscale_prio() : 0);
  }
  }  // POST_BRANCH

  hipError_t launch_err = hipSuccess;
  if (phases & POST_MAIN) {
  // --- the next draw's normals: 14 us of two fill kernels that nothing here
  // depends on, moved from the serial stretch in front of the next solve to
  // the design's stream under the tau / lambda branch (the longer one).
  // Single chains only (a batch draws its chains' normals itself).  Measured
  // (three alternating pairs, us per CG iteration): 1M x 50k 106.5 against
  // 108.2 (+1.5 %), 100k x 10k 37.9 against 37.6 (nothing: its two branches are
  // short) -- on from 250 000 rows; BBX_ETA_AHEAD=0 / 1 forces it off / on.
  static const int eta_env =
      getenv("BBX_ETA_AHEAD") ? atoi(getenv("BBX_ETA_AHEAD")) : -1;
  const bool eta_ahead = eta_env >= 0 ? eta_env == 1 : n >= 250000;
  if (eta_ahead && fork && branch_of == nullptr && (phases & POST_JOIN)) {
    if (!c->eta1_next.ptr) {
      BBX_TRY(c->eta1_next.alloc(sizeof(double) * (size_t)n));
      BBX_TRY(c->eta2_next.alloc(sizeof(double) * (size_t)P));
    }
    const uint64_t next_seed = cg_draw_seed(c, (uint64_t)c->iter + 1);
    // (with wave priority: beside the lambda kernel the n-vector fill takes
    // 49 us at normal priority and 16 us with it -- and delays the Polya-Gamma
    // kernel behind it, the longer branch by then; BBX_FILL_PRIO=0: A/B)
    static const bool fill_prio = !(getenv("BBX_FILL_PRIO") &&
                                    atoi(getenv("BBX_FILL_PRIO")) == 0);
    BBX_TRY(launch_fill_normal(h, n, next_seed, STREAM_ETA1,
                               c->eta1_next.as<double>(), fill_prio));
    BBX_TRY(launch_fill_normal(h, P, next_seed, STREAM_ETA2,
                               c->eta2_next.as<double>(), fill_prio));
    c->eta_iter = c->iter + 1;
  }
  // --- Omega | beta, continued
  int rg = grid_for(n, ROW_GRID);
  double* rp = c->row_part.as<double>();
  if (c->model == BBX_MODEL_LOGIT) {
    // elements per lane: by size; BBX_PG_ELEMS = 1 | 4 | 8 forces a width (the
    // draws do not depend on it), 0 = the one-lane kernel of rounds 1-4
    static const int pg_env =
        getenv("BBX_PG_ELEMS") ? atoi(getenv("BBX_PG_ELEMS")) : -1;
    const int elems = pg_env >= 0 ? pg_env : polya_gamma_elems(n);
#define BBX_PG_LAUNCH(KERNEL, PER_BLOCK)                                       \
  do {                                                                         \
    rg = grid_for((n + (PER_BLOCK) - 1) / (PER_BLOCK) * 256, ROW_GRID);        \
    BBX_LAUNCH(KERNEL, dim3(rg), dim3(256), 0, s, n, c->seed,          \
                       iter_stream(STREAM_PG, c->iter),                        \
                       c->outcome.as<double>(), c->n_trial.as<double>(),       \
                       c->psi.as<double>(), c->obs_prec.as<double>(), rp);     \
  } while (0)
    if (elems == 0) BBX_PG_LAUNCH(chain_pg_lane_kernel, 256);
    else if (elems >= 8) BBX_PG_LAUNCH(chain_pg_kernel<8>, 2048);
    else if (elems >= 4) BBX_PG_LAUNCH(chain_pg_kernel<4>, 1024);
    else BBX_PG_LAUNCH(chain_pg_kernel<1>, 256);
#undef BBX_PG_LAUNCH
  } else {
    BBX_LAUNCH(chain_rss_kernel, dim3(rg), dim3(256), 0, s, n,
                       c->outcome.as<double>(), c->psi.as<double>(), rp);
  }
  BBX_LAUNCH(chain_obs_finish_kernel, dim3(1), dim3(256), 0, s,
                     c->model, 0, n, c->seed,
                     iter_stream(STREAM_OBSVAR, c->iter), rp, rg, sc);

  }  // POST_MAIN

  // the second branch is joined on every path, a failed launch included
  launch_err = hipGetLastError();
  if (phases & POST_JOIN) {
    if (fork) {
      BBX_HIP(hipEventRecord(c->ev_join, s_b));
      BBX_HIP(hipStreamWaitEvent(s, c->ev_join, 0));
    }
    c->iter += 1;
  }
  BBX_HIP(launch_err);
  return BBX_OK;
}

// One Gibbs iteration (bayesbridge.py:210-240); returns the CG info (>= 0).
static int chain_step(bbx_chain* c, int maxiter, double atol, int* n_cg_iter) {
  const uint64_t it = (uint64_t)c->iter;
  BBX_TRY(chain_pre_draw(c));
  int info = 0;
  // (the normals of this draw, if the previous iteration filled them ahead)
  const bool have_eta = c->eta_iter == (long long)it && c->eta1_next.ptr;
  struct TailScope {
    bbx_design* h;
    ~TailScope() {
      h->tail_hook = nullptr;
      h->tail_ctx = nullptr;
      h->coef_copy = nullptr;
    }
  } tail_scope{c->h};
  c->h->coef_copy = c->coef_sample;
  if (chain_tail_applies(c)) {
    c->h->tail_hook = chain_tail_hook;
    c->h->tail_ctx = c;
  }

  int st = cg_sample_device(
      c->h, c->obs_prec.as<double>(), c->phi.as<double>(), c->z.as<double>(),
      c->x0.as<double>(), c->sd.as<double>(), c->n_unshrunk,
      have_eta ? c->eta1_next.as<double>() : nullptr,
      have_eta ? c->eta2_next.as<double>() : nullptr,
      cg_draw_seed(c, it), maxiter, atol, c->coef.as<double>(), n_cg_iter,
      &info, c->mean_zero ? 1 : 0);
  c->eta_iter = -1;
  if (st < 0) return st;
  c->mean_zero = false;
  // (psi is under way if the pass was enqueued at the look that found the rule fired)
  BBX_TRY(chain_post_draw(c, c->h->tail_ran, POST_ALL));
  c->h->tail_ran = false;
  return info;
}

int chain_begin_run(bbx_chain* c, int n_sample) {
  c->pending_store = -1;   // (a run that ended in an error may have left one)
  c->coef_sample = nullptr;
  BBX_TRY(c->samp_gscale.alloc(sizeof(double) * (size_t)(n_sample + 1)));
  BBX_TRY(c->samp_logp.alloc(sizeof(double) * (size_t)(n_sample + 1)));
  return BBX_OK;
}

int chain_save_sample(bbx_chain* c, int idx, double* d_coef, double* d_lscale,
                      double* d_obs_prec) {
  bbx_design* h = c->h;
  const int64_t P = h->P, n = h->n;
  const int64_t n_shrunk = P - c->n_unshrunk;
  // (coef_sample: the CG loop's finish kernel has written this sample already)
  if (d_coef && c->coef_sample != d_coef + (size_t)idx * P)
    BBX_HIP(hipMemcpyAsync(d_coef + (size_t)idx * P, c->coef.ptr,
                           sizeof(double) * (size_t)P,
                           hipMemcpyDeviceToDevice, h->stream));
  c->coef_sample = nullptr;
  if (d_lscale && n_shrunk > 0)
    BBX_HIP(hipMemcpyAsync(d_lscale + (size_t)idx * n_shrunk, c->lscale.ptr,
                           sizeof(double) * (size_t)n_shrunk,
                           hipMemcpyDeviceToDevice, h->stream));
  if (d_obs_prec) {
    if (c->model == BBX_MODEL_LOGIT)
      BBX_HIP(hipMemcpyAsync(d_obs_prec + (size_t)idx * n, c->obs_prec.ptr,
                             sizeof(double) * (size_t)n,
                             hipMemcpyDeviceToDevice, h->stream));
    else
      BBX_HIP(hipMemcpyAsync(
          d_obs_prec + idx, &c->scalars.as<ChainScalars>()->obs_prec,
          sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  }
  // global scale and log posterior: stored by the next iteration's
  // chain_prior_kernel, or by chain_end_run after the last one
  c->pending_store = idx;
  return BBX_OK;
}

int chain_end_run(bbx_chain* c, int n_sample, double* gscale, double* logp) {
  bbx_design* h = c->h;
  if (c->pending_store >= 0) {
    BBX_LAUNCH(chain_store_scalars_kernel, dim3(1), dim3(64), 0,
                       h->stream, c->scalars.as<ChainScalars>(),
                       c->pending_store, c->samp_gscale.as<double>(),
                       c->samp_logp.as<double>());
    BBX_HIP(hipGetLastError());
    c->pending_store = -1;
  }
  if (gscale && n_sample > 0)
    BBX_HIP(hipMemcpyAsync(gscale, c->samp_gscale.ptr,
                           sizeof(double) * (size_t)n_sample,
                           hipMemcpyDeviceToHost, h->stream));
  if (logp && n_sample > 0)
    BBX_HIP(hipMemcpyAsync(logp, c->samp_logp.ptr,
                           sizeof(double) * (size_t)n_sample,
                           hipMemcpyDeviceToHost, h->stream));
  BBX_HIP(hipStreamSynchronize(h->stream));
  return BBX_OK;
}

}  // namespace bbx

using namespace bbx;

extern "C" {

int bbx_chain_create(bbx_design* design, int model, const double* outcome,
                     const double* n_trial, int n_unshrunk,
                     const double* sd_unshrunk, double bridge_exp,
                     double slab_size, double gscale_shape0,
                     double gscale_rate0, uint64_t seed, bbx_chain** out) {
  if (!out) return fail(BBX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!design || !outcome) return fail(BBX_ERR_INVALID, "NULL argument");
  if (model != BBX_MODEL_LINEAR && model != BBX_MODEL_LOGIT)
    return fail(BBX_ERR_INVALID, "unknown model");
  if (n_unshrunk < 0 || n_unshrunk > design->P)
    return fail(BBX_ERR_INVALID, "n_unshrunk out of range");
  if (n_unshrunk > 0 && !sd_unshrunk)
    return fail(BBX_ERR_INVALID, "sd_unshrunk is NULL");
  if (!(bridge_exp > 0.) || bridge_exp > 2.)
    return fail(BBX_ERR_INVALID, "bridge exponent must be in (0, 2]");
  bbx_design* h = design;
  BBX_HIP(hipSetDevice(h->device));
  bbx_chain* c = new (std::nothrow) bbx_chain();
  if (!c) return fail(BBX_ERR_INVALID, "out of host memory");
  c->h = h;
  c->model = model;
  c->n_unshrunk = n_unshrunk;
  c->bridge_exp = bridge_exp;
  c->slab = slab_size;
  c->shape0 = gscale_shape0;
  c->rate0 = gscale_rate0;
  c->seed = seed;
  const int64_t n = h->n, P = h->P;
  const size_t nb = sizeof(double) * (size_t)n, Pb = sizeof(double) * (size_t)P;
  auto body = [&]() -> int {
    BBX_TRY(c->outcome.alloc(nb));
    BBX_TRY(c->n_trial.alloc(nb));
    BBX_TRY(c->kappa.alloc(nb));
    BBX_TRY(c->obs_prec.alloc(nb));
    BBX_TRY(c->psi.alloc(nb));
    for (DevMem* m : {&c->zbase, &c->coef, &c->phi, &c->x0, &c->sd, &c->z,
                      &c->mean, &c->square})
      BBX_TRY(m->alloc(Pb));
    BBX_TRY(c->sd_unshrunk.alloc(sizeof(double) * (size_t)(n_unshrunk + 1)));
    BBX_TRY(c->lscale.alloc(sizeof(double) * (size_t)(P - n_unshrunk + 1)));
    BBX_TRY(c->scalars.alloc(sizeof(ChainScalars)));
    BBX_TRY(c->row_part.alloc(sizeof(double) * ROW_GRID * 2));
    BBX_TRY(c->misc_part.alloc(sizeof(double) * NPART * 3));
    BBX_HIP(hipHostMalloc(&c->pinned, 4096, hipHostMallocDefault));
    BBX_HIP(hipMemcpy(c->outcome.ptr, outcome, nb, hipMemcpyHostToDevice));
    if (model == BBX_MODEL_LOGIT) {
      if (n_trial) {
        BBX_HIP(hipMemcpy(c->n_trial.ptr, n_trial, nb, hipMemcpyHostToDevice));
      } else {
        std::vector<double> ones((size_t)n, 1.);
        BBX_HIP(hipMemcpy(c->n_trial.ptr, ones.data(), nb,
                          hipMemcpyHostToDevice));
      }
    }
    if (n_unshrunk > 0)
      BBX_HIP(hipMemcpy(c->sd_unshrunk.ptr, sd_unshrunk,
                        sizeof(double) * (size_t)n_unshrunk,
                        hipMemcpyHostToDevice));
    // zbase = X~^T kappa (logit; bayesbridge.py:380 + reg_coef_sampler.py:74:
    // Omega cancels) or X~^T y (linear; scaled by obs_prec each iteration).
    const double* src = c->outcome.as<double>();
    if (model == BBX_MODEL_LOGIT) {
      BBX_LAUNCH(chain_kappa_kernel, dim3(grid_for(n, ROW_GRID)),
                         dim3(256), 0, h->stream, n, c->outcome.as<double>(),
                         c->n_trial.as<double>(), c->kappa.as<double>());
      src = c->kappa.as<double>();
    }
    BBX_TRY(launch_sum_n(h, src, n, part_slot(h, PS_SUMW)));
    TdotEpilogue ep;
    BBX_TRY(launch_tdot(h, src, part_slot(h, PS_SUMW), ep,
                        c->zbase.as<double>()));
    // summariser initial state (reg_coef_posterior_summarizer.py:88-91)
    BBX_HIP(hipMemsetAsync(c->mean.ptr, 0, Pb, h->stream));
    std::vector<double> ones((size_t)P, 1.);
    BBX_HIP(hipMemcpyAsync(c->square.ptr, ones.data(), Pb,
                           hipMemcpyHostToDevice, h->stream));
    BBX_HIP(hipMemsetAsync(c->coef.ptr, 0, Pb, h->stream));
    BBX_HIP(hipMemcpyAsync(c->lscale.ptr, ones.data(),
                           sizeof(double) * (size_t)(P - n_unshrunk),
                           hipMemcpyHostToDevice, h->stream));
    ChainScalars init{};
    init.gscale = 1.;
    init.obs_prec = 1.;
    BBX_HIP(hipMemcpyAsync(c->scalars.ptr, &init, sizeof(init),
                           hipMemcpyHostToDevice, h->stream));
    // Every state vector is DEFINED from here on: a chain that is run without
    // bbx_chain_set_state / bbx_chain_init_obs_prec starts from coef = 0, unit
    // scales and, for the logit model, the Polya-Gamma mean at that coefficient
    // (psi = 0: logistic_model.py:80-87) -- not from whatever the allocator
    // returned (non-finite Omega -> "non-finite residual inside CG").
    for (DevMem* m : {&c->phi, &c->x0, &c->sd, &c->z})
      BBX_HIP(hipMemsetAsync(m->ptr, 0, Pb, h->stream));
    BBX_HIP(hipMemsetAsync(c->psi.ptr, 0, nb, h->stream));
    if (model == BBX_MODEL_LOGIT)
      BBX_LAUNCH(chain_pg_mean_kernel, dim3(grid_for(n, ROW_GRID)),
                         dim3(256), 0, h->stream, n, c->n_trial.as<double>(),
                         c->psi.as<double>(), c->obs_prec.as<double>());
    else
      BBX_HIP(hipMemsetAsync(c->obs_prec.ptr, 0, nb, h->stream));
    BBX_HIP(hipGetLastError());
    BBX_HIP(hipStreamSynchronize(h->stream));
    return BBX_OK;
  };
  int st = no_throw(body);
  if (st < 0) {
    bbx_chain_destroy(c);
    return st;
  }
  *out = c;
  return BBX_OK;
}

int bbx_chain_destroy(bbx_chain* c) {
  if (!c) return BBX_OK;
  // (the design may have been destroyed first -- a garbage collector finalises
  // in any order -- and its stream with it: nothing of the chain is in flight)
  if (c->h && design_alive(c->h)) {
    (void)hipSetDevice(c->h->device);
    (void)hipStreamSynchronize(c->h->stream);
  }
  if (c->stream2) {
    (void)hipStreamSynchronize(c->stream2);
    (void)hipStreamDestroy(c->stream2);
  }
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->pinned) (void)hipHostFree(c->pinned);
  delete c;
  return BBX_OK;
}

static int bbx_chain_set_state_impl(bbx_chain* c, const double* coef,
                        const double* obs_prec, const double* lscale,
                        const double* gscale) {
  BBX_TRY(chain_check(c));
  bbx_design* h = c->h;
  BBX_HIP(hipSetDevice(h->device));
  if (coef)
    BBX_HIP(hipMemcpy(c->coef.ptr, coef, sizeof(double) * (size_t)h->P,
                      hipMemcpyHostToDevice));
  ChainScalars sc;
  BBX_HIP(hipMemcpy(&sc, c->scalars.ptr, sizeof(sc), hipMemcpyDeviceToHost));
  if (obs_prec) {
    if (c->model == BBX_MODEL_LOGIT)
      BBX_HIP(hipMemcpy(c->obs_prec.ptr, obs_prec,
                        sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice));
    else
      sc.obs_prec = obs_prec[0];
  }
  if (lscale && h->P > c->n_unshrunk)
    BBX_HIP(hipMemcpy(c->lscale.ptr, lscale,
                      sizeof(double) * (size_t)(h->P - c->n_unshrunk),
                      hipMemcpyHostToDevice));
  if (gscale) sc.gscale = *gscale;
  BBX_HIP(hipMemcpy(c->scalars.ptr, &sc, sizeof(sc), hipMemcpyHostToDevice));
  return BBX_OK;
}

int bbx_chain_set_state(bbx_chain* c, const double* coef,
                        const double* obs_prec, const double* lscale,
                        const double* gscale) {
  return no_throw([&]() -> int {
    return bbx_chain_set_state_impl(c, coef, obs_prec, lscale, gscale);
  });
}


static int bbx_chain_get_state_impl(bbx_chain* c, double* coef, double* obs_prec,
                        double* lscale, double* gscale) {
  BBX_TRY(chain_check(c));
  bbx_design* h = c->h;
  BBX_HIP(hipSetDevice(h->device));
  BBX_HIP(hipStreamSynchronize(h->stream));
  if (coef)
    BBX_HIP(hipMemcpy(coef, c->coef.ptr, sizeof(double) * (size_t)h->P,
                      hipMemcpyDeviceToHost));
  ChainScalars sc;
  BBX_HIP(hipMemcpy(&sc, c->scalars.ptr, sizeof(sc), hipMemcpyDeviceToHost));
  if (obs_prec) {
    if (c->model == BBX_MODEL_LOGIT)
      BBX_HIP(hipMemcpy(obs_prec, c->obs_prec.ptr,
                        sizeof(double) * (size_t)h->n, hipMemcpyDeviceToHost));
    else
      obs_prec[0] = sc.obs_prec;
  }
  if (lscale && h->P > c->n_unshrunk)
    BBX_HIP(hipMemcpy(lscale, c->lscale.ptr,
                      sizeof(double) * (size_t)(h->P - c->n_unshrunk),
                      hipMemcpyDeviceToHost));
  if (gscale) *gscale = sc.gscale;
  return BBX_OK;
}

int bbx_chain_get_state(bbx_chain* c, double* coef, double* obs_prec,
                        double* lscale, double* gscale) {
  return no_throw([&]() -> int {
    return bbx_chain_get_state_impl(c, coef, obs_prec, lscale, gscale);
  });
}


int bbx_chain_set_summary(bbx_chain* c, const double* mean,
                          const double* square, int64_t n_averaged) {
  BBX_TRY(chain_check(c));
  if (!mean || !square || n_averaged < 0)
    return fail(BBX_ERR_INVALID, "bad summary");
  BBX_HIP(hipSetDevice(c->h->device));
  const size_t Pb = sizeof(double) * (size_t)c->h->P;
  BBX_HIP(hipMemcpy(c->mean.ptr, mean, Pb, hipMemcpyHostToDevice));
  BBX_HIP(hipMemcpy(c->square.ptr, square, Pb, hipMemcpyHostToDevice));
  c->n_averaged = n_averaged;
  c->mean_zero = true;
  for (int64_t j = 0; j < c->h->P; ++j)
    if (mean[j] != 0.) {
      c->mean_zero = false;
      break;
    }
  return BBX_OK;
}

int bbx_chain_get_summary(bbx_chain* c, double* mean, double* square,
                          int64_t* n_averaged) {
  BBX_TRY(chain_check(c));
  BBX_HIP(hipSetDevice(c->h->device));
  BBX_HIP(hipStreamSynchronize(c->h->stream));
  const size_t Pb = sizeof(double) * (size_t)c->h->P;
  if (mean) BBX_HIP(hipMemcpy(mean, c->mean.ptr, Pb, hipMemcpyDeviceToHost));
  if (square)
    BBX_HIP(hipMemcpy(square, c->square.ptr, Pb, hipMemcpyDeviceToHost));
  if (n_averaged) *n_averaged = c->n_averaged;
  return BBX_OK;
}

int bbx_chain_get_iteration(bbx_chain* c, int64_t* iteration) {
  BBX_TRY(chain_check(c));
  if (iteration) *iteration = c->iter;
  return BBX_OK;
}

int bbx_chain_set_iteration(bbx_chain* c, int64_t iteration) {
  BBX_TRY(chain_check(c));
  if (iteration < 0) return fail(BBX_ERR_INVALID, "iteration < 0");
  c->iter = iteration;
  c->eta_iter = -1;   // (normals filled ahead belong to the old numbering)
  return BBX_OK;
}

int bbx_chain_get_seed(bbx_chain* c, uint64_t* seed) {
  BBX_TRY(chain_check(c));
  if (seed) *seed = c->seed;
  return BBX_OK;
}

int bbx_chain_set_seed(bbx_chain* c, uint64_t seed) {
  BBX_TRY(chain_check(c));
  c->seed = seed;
  c->eta_iter = -1;   // (normals filled ahead used the old seed)
  return BBX_OK;
}

int bbx_chain_set_gscale_update(bbx_chain* c, int mode) {
  BBX_TRY(chain_check(c));
  if (mode != BBX_GSCALE_SAMPLE && mode != BBX_GSCALE_OPTIMIZE &&
      mode != BBX_GSCALE_FIXED)
    return fail(BBX_ERR_INVALID, "unknown global-scale update mode");
  c->gscale_update = mode;
  return BBX_OK;
}

static int bbx_chain_eta_impl(bbx_chain* c, int64_t iteration, double* eta1,
                              double* eta2) {
  BBX_TRY(chain_check(c));
  if (iteration < 0) return fail(BBX_ERR_INVALID, "iteration < 0");
  bbx_design* h = c->h;
  BBX_HIP(hipSetDevice(h->device));
  const uint64_t seed = cg_draw_seed(c, (uint64_t)iteration);
  DevMem tmp;
  const int64_t len = h->n > h->P ? h->n : h->P;
  BBX_TRY(tmp.alloc(sizeof(double) * (size_t)len));
  if (eta1) {
    BBX_TRY(launch_fill_normal(h, h->n, seed, STREAM_ETA1, tmp.as<double>()));
    BBX_HIP(hipMemcpyAsync(eta1, tmp.ptr, sizeof(double) * (size_t)h->n,
                           hipMemcpyDeviceToHost, h->stream));
    BBX_HIP(hipStreamSynchronize(h->stream));
  }
  if (eta2) {
    BBX_TRY(launch_fill_normal(h, h->P, seed, STREAM_ETA2, tmp.as<double>()));
    BBX_HIP(hipMemcpyAsync(eta2, tmp.ptr, sizeof(double) * (size_t)h->P,
                           hipMemcpyDeviceToHost, h->stream));
    BBX_HIP(hipStreamSynchronize(h->stream));
  }
  return BBX_OK;
}

int bbx_chain_eta(bbx_chain* c, int64_t iteration, double* eta1,
                  double* eta2) {
  return no_throw([&]() -> int {
    return bbx_chain_eta_impl(c, iteration, eta1, eta2);
  });
}

int bbx_chain_get_logp(bbx_chain* c, double* loglik, double* logp) {
  BBX_TRY(chain_check(c));
  BBX_HIP(hipSetDevice(c->h->device));
  BBX_HIP(hipStreamSynchronize(c->h->stream));
  ChainScalars sc;
  BBX_HIP(hipMemcpy(&sc, c->scalars.ptr, sizeof(sc), hipMemcpyDeviceToHost));
  if (loglik) *loglik = sc.loglik;
  if (logp) *logp = sc.logp();
  return BBX_OK;
}

int bbx_chain_init_obs_prec(bbx_chain* c) {
  BBX_TRY(chain_check(c));
  bbx_design* h = c->h;
  BBX_HIP(hipSetDevice(h->device));
  BBX_TRY(chain_linear_predictor(c));
  const int rg = grid_for(h->n, ROW_GRID);
  if (c->model == BBX_MODEL_LOGIT) {
    BBX_LAUNCH(chain_pg_mean_kernel, dim3(rg), dim3(256), 0, h->stream,
                       h->n, c->n_trial.as<double>(), c->psi.as<double>(),
                       c->obs_prec.as<double>());
  } else {
    double* rp = c->row_part.as<double>();
    BBX_LAUNCH(chain_rss_kernel, dim3(rg), dim3(256), 0, h->stream,
                       h->n, c->outcome.as<double>(), c->psi.as<double>(), rp);
    BBX_LAUNCH(chain_obs_finish_kernel, dim3(1), dim3(256), 0,
                       h->stream, c->model, 1, h->n, c->seed, 0, rp, rg,
                       c->scalars.as<ChainScalars>());
  }
  BBX_HIP(hipGetLastError());
  BBX_HIP(hipStreamSynchronize(h->stream));
  return BBX_OK;
}

static int bbx_chain_run_impl(bbx_chain* c, int n_iter, int n_burnin, int thin,
                  int maxiter, double atol, double* d_coef, double* d_lscale,
                  double* d_obs_prec, double* gscale, double* logp,
                  double* n_cg_iter) {
  BBX_TRY(chain_check(c));
  if (n_iter < 0 || n_burnin < 0 || thin < 1 || n_burnin > n_iter)
    return fail(BBX_ERR_INVALID, "bad n_iter / n_burnin / thin");
  bbx_design* h = c->h;
  BBX_HIP(hipSetDevice(h->device));
  const int64_t P = h->P;
  if (maxiter <= 0) maxiter = 500;                      // reg_coef_sampler.py:95
  if (!(atol > 0.)) atol = 10e-6 * std::sqrt((double)P);
  const int n_sample = (n_iter - n_burnin) / thin;
  BBX_TRY(chain_begin_run(c, n_sample));
  int n_unconverged = 0;
  for (int it = 1; it <= n_iter; ++it) {
    int ncg = 0;
    const bool kept = it > n_burnin && (it - n_burnin) % thin == 0 &&
                      (it - n_burnin) / thin - 1 < n_sample;
    const int idx = kept ? (it - n_burnin) / thin - 1 : -1;  // gibbs_util.py:170
    // a kept iteration's coefficients go to their sample slot straight from the
    // CG loop's finish kernel (one D2D copy less on the stretch between solves)
    c->coef_sample = (kept && d_coef) ? d_coef + (size_t)idx * P : nullptr;
    int info = chain_step(c, maxiter, atol, &ncg);
    if (info < 0) return info;
    if (info > 0) ++n_unconverged;
    if (kept) {
      BBX_TRY(chain_save_sample(c, idx, d_coef, d_lscale, d_obs_prec));
      if (n_cg_iter) n_cg_iter[idx] = (double)ncg;
    }
    // (the draw of iteration `it` is complete on the host's side of the
    // queue: the CG loop has seen its stop test fire)
    if (c->progress && c->progress_every > 0 && it % c->progress_every == 0)
      c->progress(it, c->progress_ctx);
  }
  BBX_TRY(chain_end_run(c, n_sample, gscale, logp));
  return n_unconverged;
}

int bbx_chain_set_progress(bbx_chain* c, int every, void (*fn)(int, void*),
                           void* ctx) {
  BBX_TRY(chain_check(c));
  if (every < 0) return fail(BBX_ERR_INVALID, "every must be >= 0");
  c->progress = (every > 0) ? fn : nullptr;
  c->progress_ctx = ctx;
  c->progress_every = fn ? every : 0;
  return BBX_OK;
}

int bbx_chain_run(bbx_chain* c, int n_iter, int n_burnin, int thin,
                  int maxiter, double atol, double* d_coef, double* d_lscale,
                  double* d_obs_prec, double* gscale, double* logp,
                  double* n_cg_iter) {
  return no_throw([&]() -> int {
    return bbx_chain_run_impl(c, n_iter, n_burnin, thin, maxiter, atol, d_coef, d_lscale, d_obs_prec, gscale, logp, n_cg_iter);
  });
}


static int bbx_chain_run_host_impl(bbx_chain* c, int n_iter, int n_burnin, int thin,
                       int maxiter, double atol, double* coef, double* lscale,
                       double* obs_prec, double* gscale, double* logp,
                       double* n_cg_iter) {
  BBX_TRY(chain_check(c));
  if (n_iter < 0 || n_burnin < 0 || thin < 1 || n_burnin > n_iter)
    return fail(BBX_ERR_INVALID, "bad n_iter / n_burnin / thin");
  bbx_design* h = c->h;
  BBX_HIP(hipSetDevice(h->device));
  const int64_t P = h->P, n = h->n;
  const int64_t n_sample = (n_iter - n_burnin) / thin;
  const int64_t n_shrunk = P - c->n_unshrunk;
  const int64_t op_len = (c->model == BBX_MODEL_LOGIT) ? n : 1;
  DevMem dc, dl, dp;
  if (coef) BBX_TRY(dc.alloc(sizeof(double) * (size_t)(n_sample * P + 1)));
  if (lscale)
    BBX_TRY(dl.alloc(sizeof(double) * (size_t)(n_sample * n_shrunk + 1)));
  if (obs_prec)
    BBX_TRY(dp.alloc(sizeof(double) * (size_t)(n_sample * op_len + 1)));
  int st = bbx_chain_run(c, n_iter, n_burnin, thin, maxiter, atol,
                         coef ? dc.as<double>() : nullptr,
                         lscale ? dl.as<double>() : nullptr,
                         obs_prec ? dp.as<double>() : nullptr, gscale, logp,
                         n_cg_iter);
  if (st < 0) return st;
  if (coef && n_sample > 0)
    BBX_HIP(hipMemcpy(coef, dc.ptr, sizeof(double) * (size_t)(n_sample * P),
                      hipMemcpyDeviceToHost));
  if (lscale && n_sample * n_shrunk > 0)
    BBX_HIP(hipMemcpy(lscale, dl.ptr,
                      sizeof(double) * (size_t)(n_sample * n_shrunk),
                      hipMemcpyDeviceToHost));
  if (obs_prec && n_sample > 0)
    BBX_HIP(hipMemcpy(obs_prec, dp.ptr,
                      sizeof(double) * (size_t)(n_sample * op_len),
                      hipMemcpyDeviceToHost));
  return st;
}

int bbx_chain_run_host(bbx_chain* c, int n_iter, int n_burnin, int thin,
                       int maxiter, double atol, double* coef, double* lscale,
                       double* obs_prec, double* gscale, double* logp,
                       double* n_cg_iter) {
  return no_throw([&]() -> int {
    return bbx_chain_run_host_impl(c, n_iter, n_burnin, thin, maxiter, atol, coef, lscale, obs_prec, gscale, logp, n_cg_iter);
  });
}


// ---- stand-alone device samplers (distribution tests)

static int dev_sampler_common(int device, int64_t n_draw) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(BBX_ERR_NODEVICE, "no HIP device visible");
  if (device < 0 || device >= count)
    return fail(BBX_ERR_INVALID, "device index out of range");
  if (n_draw < 0) return fail(BBX_ERR_INVALID, "n_draw < 0");
  BBX_HIP(hipSetDevice(device));
  return BBX_OK;
}

static int bbx_device_polya_gamma_impl(int device, uint64_t seed, int64_t n_draw,
                           const int32_t* shape, const double* tilt,
                           double* out) {
  BBX_TRY(dev_sampler_common(device, n_draw));
  if (n_draw == 0) return BBX_OK;
  if (!shape || !tilt || !out) return fail(BBX_ERR_INVALID, "NULL argument");
  DevMem ds, dt, dout;
  BBX_TRY(ds.alloc(sizeof(int32_t) * (size_t)n_draw));
  BBX_TRY(dt.alloc(sizeof(double) * (size_t)n_draw));
  BBX_TRY(dout.alloc(sizeof(double) * (size_t)n_draw));
  BBX_HIP(hipMemcpy(ds.ptr, shape, sizeof(int32_t) * (size_t)n_draw,
                    hipMemcpyHostToDevice));
  BBX_HIP(hipMemcpy(dt.ptr, tilt, sizeof(double) * (size_t)n_draw,
                    hipMemcpyHostToDevice));
  // (the draws do not depend on the elements per lane: the small width for
  // short vectors, the chain's widest otherwise)
  if (n_draw < 50000)
    BBX_LAUNCH(dev_pg_kernel<1>, dim3(grid_for(n_draw, 4096)), dim3(256),
                       0, 0, n_draw, seed, ds.as<int32_t>(), dt.as<double>(),
                       dout.as<double>());
  else
    BBX_LAUNCH(dev_pg_kernel<8>, dim3(grid_for((n_draw + 7) / 8, 4096)),
                       dim3(256), 0, 0, n_draw, seed, ds.as<int32_t>(),
                       dt.as<double>(), dout.as<double>());
  BBX_HIP(hipGetLastError());
  BBX_HIP(hipMemcpy(out, dout.ptr, sizeof(double) * (size_t)n_draw,
                    hipMemcpyDeviceToHost));
  return BBX_OK;
}

int bbx_device_polya_gamma(int device, uint64_t seed, int64_t n_draw,
                           const int32_t* shape, const double* tilt,
                           double* out) {
  return no_throw([&]() -> int {
    return bbx_device_polya_gamma_impl(device, seed, n_draw, shape, tilt, out);
  });
}


static int bbx_device_tilted_stable_impl(int device, uint64_t seed, int64_t n_draw,
                             double char_exp, const double* tilt,
                             double* out) {
  BBX_TRY(dev_sampler_common(device, n_draw));
  if (n_draw == 0) return BBX_OK;
  if (!tilt || !out) return fail(BBX_ERR_INVALID, "NULL argument");
  if (!(char_exp > 0.) || !(char_exp < 1.))
    return fail(BBX_ERR_INVALID, "characteristic exponent must be in (0,1)");
  DevMem dt, dout;
  BBX_TRY(dt.alloc(sizeof(double) * (size_t)n_draw));
  BBX_TRY(dout.alloc(sizeof(double) * (size_t)n_draw));
  BBX_HIP(hipMemcpy(dt.ptr, tilt, sizeof(double) * (size_t)n_draw,
                    hipMemcpyHostToDevice));
  BBX_LAUNCH(dev_ts_kernel, dim3(grid_for(n_draw, 4096)), dim3(256), 0,
                     0, n_draw, seed, char_exp, dt.as<double>(),
                     dout.as<double>(), bbx::ts_cost_threshold());
  BBX_HIP(hipGetLastError());
  BBX_HIP(hipMemcpy(out, dout.ptr, sizeof(double) * (size_t)n_draw,
                    hipMemcpyDeviceToHost));
  return BBX_OK;
}

int bbx_device_tilted_stable(int device, uint64_t seed, int64_t n_draw,
                             double char_exp, const double* tilt,
                             double* out) {
  return no_throw([&]() -> int {
    return bbx_device_tilted_stable_impl(device, seed, n_draw, char_exp, tilt, out);
  });
}


static int bbx_device_gamma_impl(int device, uint64_t seed, int64_t n_draw, double shape,
                     double* out) {
  BBX_TRY(dev_sampler_common(device, n_draw));
  if (n_draw == 0) return BBX_OK;
  if (!out || !(shape > 0.)) return fail(BBX_ERR_INVALID, "bad argument");
  DevMem dout;
  BBX_TRY(dout.alloc(sizeof(double) * (size_t)n_draw));
  BBX_LAUNCH(dev_gamma_kernel, dim3(grid_for(n_draw, 4096)), dim3(256),
                     0, 0, n_draw, seed, shape, dout.as<double>());
  BBX_HIP(hipGetLastError());
  BBX_HIP(hipMemcpy(out, dout.ptr, sizeof(double) * (size_t)n_draw,
                    hipMemcpyDeviceToHost));
  return BBX_OK;
}

int bbx_device_gamma(int device, uint64_t seed, int64_t n_draw, double shape,
                     double* out) {
  return no_throw([&]() -> int {
    return bbx_device_gamma_impl(device, seed, n_draw, shape, out);
  });
}


__global__ __launch_bounds__(256) void dev_normal_kernel(
    int64_t n, uint64_t seed, uint64_t stream, double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256) {
    bbx::Philox g(seed, stream, (uint64_t)i);
    out[i] = g.normal();
  }
}

static int bbx_device_normal_impl(int device, uint64_t seed, uint64_t stream,
                                  int64_t n_draw, double* out) {
  BBX_TRY(dev_sampler_common(device, n_draw));
  if (n_draw == 0) return BBX_OK;
  if (!out) return fail(BBX_ERR_INVALID, "out is NULL");
  DevMem dout;
  BBX_TRY(dout.alloc(sizeof(double) * (size_t)n_draw));
  BBX_LAUNCH(dev_normal_kernel, dim3(grid_for(n_draw, 4096)),
                     dim3(256), 0, 0, n_draw, seed, stream, dout.as<double>());
  BBX_HIP(hipGetLastError());
  BBX_HIP(hipMemcpy(out, dout.ptr, sizeof(double) * (size_t)n_draw,
                    hipMemcpyDeviceToHost));
  return BBX_OK;
}

int bbx_device_normal(int device, uint64_t seed, uint64_t stream,
                      int64_t n_draw, double* out) {
  return no_throw([&]() -> int {
    return bbx_device_normal_impl(device, seed, stream, n_draw, out);
  });
}

}  // extern "C"
