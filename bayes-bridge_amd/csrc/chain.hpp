// Device-resident Gibbs chain: the handle and the three phases of one
// iteration (bayesbridge.py:210-240), shared by chain.hip (one chain per pass
// over the matrix) and batch.hip (K chains per pass).
#pragma once
#include <cmath>

#include "common.hpp"

namespace bbx {

constexpr int ROW_GRID = 2048;  // blocks of the n-length sampler kernels

// Two 128-byte halves: after the coefficient draw the Omega update and the
// tau / lambda updates run as two branches on two streams (chain_step), each
// writing only its own half, so that no cache line is dirty in two L2s.
struct alignas(128) ChainScalars {
  // ---- written by the tau / lambda branch
  double gscale;         // tau, raw parametrisation
  double logprior;       // log posterior minus the log-likelihood
  double abs_pow_sum;    // sum |beta_j|^alpha over shrunk coordinates
  long long n_gscale_clamped;
  long long n_lscale_fixed;
  // ---- written by the Omega branch
  alignas(128) double obs_prec;  // linear model only
  double loglik;                 // of the current coef
  // log posterior (bayesbridge.py:480-511)
  __host__ __device__ double logp() const { return loglik + logprior; }
};

}  // namespace bbx

struct bbx_chain {
  bbx_design* h = nullptr;
  int model = BBX_MODEL_LOGIT;
  int n_unshrunk = 0;
  double bridge_exp = .5, slab = INFINITY, shape0 = 0., rate0 = 0.;
  uint64_t seed = 0;
  int64_t iter = 0;        // iterations done (Philox key)
  int64_t n_averaged = 0;  // summariser count
  bool mean_zero = true;   // running mean still all zeros => CG warm start 0
  int gscale_update = BBX_GSCALE_SAMPLE;
  // bbx_chain_set_progress: called from the host loop of a run every
  // `progress_every` iterations (gibbs_util.py:214-238 prints from there)
  void (*progress)(int, void*) = nullptr;
  void* progress_ctx = nullptr;
  int progress_every = 0;
  bbx::DevMem outcome, n_trial, kappa;  // n
  bbx::DevMem zbase;                    // P: X~^T kappa (logit) or X~^T y
  bbx::DevMem coef, phi, x0, sd, z, mean, square, sd_unshrunk;  // P-length
  bbx::DevMem lscale;                   // P - n_unshrunk
  bbx::DevMem obs_prec, psi;            // n
  bbx::DevMem scalars;                  // ChainScalars
  bbx::DevMem row_part;                 // ROW_GRID partials x 2
  bbx::DevMem misc_part;                // NPART x 3: sums over the coefficients (tau branch)
  // The normals of the NEXT draw (cg_sampler.py:61-62), filled on the design's
  // stream while the Polya-Gamma and lambda kernels run: Philox values depend on
  // (seed, iteration, element) only, so WHEN they are generated changes no bit.
  // eta_iter = the iteration these buffers hold (-1: none).
  bbx::DevMem eta1_next, eta2_next;     // n, P
  long long eta_iter = -1;
  bbx::DevMem samp_gscale, samp_logp;   // per kept sample (device)
  // deferred bookkeeping of a kept iteration: its scalars are stored by the
  // next chain_prior_kernel (pending_store = sample index, -1: none); its
  // coefficients by the CG loop's finish kernel when the run loop announced the
  // slot beforehand (coef_sample, else chain_save_sample copies)
  int pending_store = -1;
  double* coef_sample = nullptr;
  void* pinned = nullptr;
  // second stream for the tau / lambda branch of an iteration
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_join = nullptr;
};

namespace bbx {

// Philox key of the CG draw's normals at 0-based iteration `it`.
inline uint64_t cg_draw_seed(const bbx_chain* c, uint64_t it) {
  return c->seed + 0x9E3779B97F4A7C15ull * (it + 1);
}

// beta | Omega, tau, lambda, part 1 (bayesbridge.py:372-395): phi, the CG warm
// start, the preconditioner sd and z of this iteration, on the design's stream.
int chain_pre_draw(bbx_chain* c);
// Everything after the coefficient draw: Omega | beta (the linear predictor
// psi = X~ beta unless `have_psi`: a batch computes it for all its chains in
// one pass), running summaries, tau | beta, lambda | tau, beta, log posterior.
// In three phases, so that a batch can start the tau / lambda branch of EVERY
// chain (each on the chain's own second stream, with the chain's own partial
// slots) before the first Omega update occupies the design's stream:
// POST_BRANCH launches the branch, POST_MAIN the Omega update on the design's
// stream, POST_JOIN makes the design's stream wait for the branch and advances
// c->iter.  One chain on its own: POST_ALL.  `branch_of`: the chain whose second
// stream carries this chain's branch (a wide batch folds its chains' branches
// onto a few streams); nullptr = the chain's own.
enum { POST_BRANCH = 1, POST_MAIN = 2, POST_JOIN = 4, POST_ALL = 7 };
int chain_post_draw(bbx_chain* c, bool have_psi, int phases,
                    bbx_chain* branch_of = nullptr);
// Sample bookkeeping of a run (gibbs_util.py:126-174): per-sample scalars are
// collected on the device and copied out once at the end.
int chain_begin_run(bbx_chain* c, int n_sample);
int chain_save_sample(bbx_chain* c, int idx, double* d_coef, double* d_lscale,
                      double* d_obs_prec);
int chain_end_run(bbx_chain* c, int n_sample, double* gscale, double* logp);

}  // namespace bbx
