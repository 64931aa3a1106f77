// Batched chains: K independent Gibbs chains on one GPU share every pass over
// the design matrix.
//
// The reference runs one chain per process (bayesbridge.py:109); its hot loop
// is the operator of cg_sampler.py:105-108, whose cost is the stream of the
// matrix.  That stream does not depend on the chain -- only Omega, s, d and
// the vectors do -- so K chains can ride on ONE pass: the sparse products
// become K-column products over the same id stream (spmv_tiled.hip, KP > 0:
// K interleaved vector slices in LDS, a ds_read_b128 serves two chains), the
// dense operator a K-column pass (dense.hip).  Everything else of an iteration
// stays per chain and is the code of chain.hip (same kernels, same Philox
// keys), and every column of a batched kernel performs exactly the additions,
// in the order, that it performs for any other content of the neighbouring
// columns: a chain's samples do not depend on which chains it is batched with
// (tests/test_hip_batch.py compares bit for bit).
//
// Layout: the CG vectors of a batch are interleaved, v[j * K + c] (one 16-byte
// access carries element j of two chains; the products want exactly this);
// what belongs to a chain (phi, z, x0, sd, Omega, eta, coef, psi) stays in the
// chain's own arrays and reaches the kernels as per-chain pointers.
// Every chain has its own CGState and stops on its own; the batch's loop runs
// until the last one has stopped (a finished chain's column idles: its x and
// r are no longer touched, the id stream is read once either way).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <new>
#include <vector>

#include "chain.hpp"
#include "philox.hpp"

namespace bbx {

struct BatchState {
  CGState st[BATCH_MAX];
  int n_done;    // chains whose stop rule has fired
  int all_done;  // n_done == K: the operator kernels exit at entry
  int pad[2];
};

}  // namespace bbx

struct bbx_batch {
  bbx_design* h = nullptr;
  int K = 0;   // chains
  int KS = 0;  // interleave stride of the batch's vectors: K (sparse), 16 or 32 (dense)
  bbx_chain* chain[bbx::BATCH_MAX] = {};
  bbx::DevMem s, d, x, r, p, sp;  // (P + 2) * KS doubles, interleaved [j][c]
  bbx::DevMem t, w;               // n * KS doubles (dense: zero-padded rows too)
  bbx::DevMem eta1[bbx::BATCH_MAX], eta2[bbx::BATCH_MAX];  // n, P per chain
  bbx::DevMem part;               // [PS_COUNT][K][NPART]
  bbx::DevMem state;              // BatchState
  void* pinned = nullptr;
  int last_cg_iter = 0;
  int n_unconverged[bbx::BATCH_MAX] = {};  // per chain, of the last run
};

namespace bbx {

static inline double* bpart(const bbx_batch* b, int slot) {
  return b->part.as<double>() + (size_t)slot * b->K * NPART;
}

// ------------------------------------------------------------------ kernels
//
// The vector kernels of a batch handle KC = min(chains, 4) chains per thread
// (their elements of the interleaved vectors are adjacent) and chains / KC
// groups of chains in blockIdx.y; `c0` is the group's first chain.  Their
// parameter `K` is the interleave STRIDE of the vectors (the number of chains
// for sparse designs, 16 for dense ones).

// the sum of one NPART-block, same adds in the same order in every thread
// (vecops.hip part_issue / part_finish, without the LDS round)
struct PartLoadK {
  double v[NPART / WAVE];
};
__device__ inline PartLoadK part_issue_k(const double* part) {
  PartLoadK p;
#pragma unroll
  for (int k = 0; k < NPART / WAVE; ++k)
    p.v[k] = part[(threadIdx.x & (WAVE - 1)) + k * WAVE];
  return p;
}
__device__ inline double part_finish_k(const PartLoadK& p) {
  double a = 0.;
#pragma unroll
  for (int k = 0; k < NPART / WAVE; ++k) a += p.v[k];
  return wave_allsum(a);  // every wave computes the same value
}

// per-chain block partial: part[(c0 + c) * NPART + blockIdx.x]
template <int KC>
__device__ inline void block_store_partials_k(const double (&x)[KC],
                                              double* part, int c0) {
  __shared__ double s_w[KC][VEC_BLOCK / WAVE];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const double v = wave_allsum(x[c]);
    if ((threadIdx.x & (WAVE - 1)) == 0) s_w[c][threadIdx.x / WAVE] = v;
  }
  __syncthreads();
  if (threadIdx.x < KC) {
    double r = 0.;
#pragma unroll
    for (int k = 0; k < VEC_BLOCK / WAVE; ++k) r += s_w[threadIdx.x][k];
    part[(c0 + (int)threadIdx.x) * NPART + blockIdx.x] = r;
  }
  __syncthreads();
}

// s, d and the scaled warm start of every chain (cg_sampler.py:104,128-138,76)
template <int KC>
__global__ __launch_bounds__(VEC_BLOCK) void b_setup_kernel(
    int64_t P, int K, int n_unshrunk, ChainPtrs phi, ChainPtrs sd, ChainPtrs x0,
    double* __restrict__ s, double* __restrict__ d, double* __restrict__ xs,
    BatchState* __restrict__ bs, double atol) {
  const int c0 = blockIdx.y * KC;
  if (blockIdx.x == 0 && threadIdx.x < KC) {
    CGState* st = &bs->st[c0 + threadIdx.x];
    st->rho[0] = st->rho[1] = 0.;
    st->atol = atol;
    st->bnorm2 = 0.;
    st->n_iter = 0;
    st->done = 0;
    st->bad = 0;
    st->pad = 0;
    if (threadIdx.x == 0 && blockIdx.y == 0) {
      bs->n_done = 0;
      bs->all_done = 0;
    }
  }
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK) {
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const double ph = phi.p[c0 + c][jj];
      const double sj = (jj < n_unshrunk) ? 2. * sd.p[c0 + c][jj] : 1. / ph;
      const double sp = sj * ph;
      s[jj * K + c0 + c] = sj;
      d[jj * K + c0 + c] = sp * sp;
      xs[jj * K + c0 + c] = x0.p[c0 + c][jj] / sj;
    }
  }
}

// v_c = s_c .* x_c (or x_c) into the interleaved buffer, and the partials of
// <offset, v_c[1:]>.  x comes interleaved (x_il) or from per-chain arrays.
template <int KC>
__global__ __launch_bounds__(VEC_BLOCK) void b_prep_kernel(
    int64_t P, int K, int intercept, const double* __restrict__ x_il,
    ChainPtrs x_sep, const double* __restrict__ s_il,
    const double* __restrict__ offset, double* __restrict__ v,
    double* __restrict__ c_part) {
  const int c0 = blockIdx.y * KC;
  double acc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) acc[c] = 0.;
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK) {
    const double off = jj >= intercept ? offset[jj - intercept] : 0.;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      double val = x_il ? x_il[jj * K + c0 + c] : x_sep.p[c0 + c][jj];
      if (s_il) val *= s_il[jj * K + c0 + c];
      v[jj * K + c0 + c] = val;
      if (jj >= intercept) acc[c] += off * val;
    }
  }
  block_store_partials_k<KC>(acc, c_part, c0);
}

// w_c = minus_c - sqrt(Omega_c) eta1_c (minus == nullptr: -sqrt(Omega_c) eta1_c)
// interleaved, and the partials of sum(w_c).
template <int KC>
__global__ __launch_bounds__(VEC_BLOCK) void b_sqrt_scale_kernel(
    int64_t n, int K, ChainPtrs omega, ChainPtrs eta,
    const double* __restrict__ minus, double* __restrict__ w,
    double* __restrict__ part) {
  const int c0 = blockIdx.y * KC;
  double acc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) acc[c] = 0.;
  for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * VEC_BLOCK) {
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      double val = sqrt(omega.p[c0 + c][i]) * eta.p[c0 + c][i];
      if (minus) val = minus[i * K + c0 + c] - val;
      else val = -val;
      w[i * K + c0 + c] = val;
      acc[c] += val;
    }
  }
  block_store_partials_k<KC>(acc, part, c0);
}

// Top of CG iteration k for every chain that is still running (vecops.hip
// cg_direction_kernel per column): stop test, rho, beta, p, s .* p, partials of
// <offset, (s p)[1:]> and <p, d p>.
template <int KC>
__global__ __launch_bounds__(VEC_BLOCK) void b_direction_kernel(
    int64_t P, int K, int n_chain, int intercept, int k,
    BatchState* __restrict__ bs,
    const double* __restrict__ rr_part, const double* __restrict__ r,
    double* __restrict__ pvec, const double* __restrict__ s,
    const double* __restrict__ offset, double* __restrict__ sp,
    double* __restrict__ c_part, const double* __restrict__ d,
    double* __restrict__ pdp_part) {
  const int c0 = blockIdx.y * KC;
  PartLoadK pl[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) pl[c] = part_issue_k(rr_part + (c0 + c) * NPART);
  bool run[KC];
  double beta[KC], rho[KC];
  bool any = false;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const CGState* st = &bs->st[c0 + c];
    const int was_done = st->done;
    const double atol = st->atol;
    const double rho_prev = (k > 0) ? st->rho[(k - 1) & 1] : 1.;
    rho[c] = part_finish_k(pl[c]);
    const bool finite = (rho[c] == rho[c]) && (rho[c] - rho[c] == 0.);
    const bool stop = !was_done && (!finite || sqrt(rho[c]) < atol);
    run[c] = !was_done && !stop;
    beta[c] = (k > 0) ? rho[c] / rho_prev : 0.;
    any = any || run[c];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (stop) {
        bs->st[c0 + c].done = 1;
        if (!finite) bs->st[c0 + c].bad = 1;
        // the last chain to stop raises the flag the operator kernels read
        if (atomicAdd(&bs->n_done, 1) + 1 == n_chain) bs->all_done = 1;
      } else if (run[c]) {
        bs->st[c0 + c].rho[k & 1] = rho[c];
      }
    }
  }
  if (!any) return;
  double acc[KC], acc_d[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) acc[c] = acc_d[c] = 0.;
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK) {
    const double off = jj >= intercept ? offset[jj - intercept] : 0.;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      if (!run[c]) continue;
      const int64_t at = jj * K + c0 + c;
      double pj = r[at];
      if (k > 0) pj += beta[c] * pvec[at];
      pvec[at] = pj;
      const double v = s[at] * pj;
      sp[at] = v;
      if (jj >= intercept) acc[c] += off * v;
      acc_d[c] += d[at] * pj * pj;
    }
  }
  block_store_partials_k<KC>(acc, c_part, c0);
  block_store_partials_k<KC>(acc_d, pdp_part, c0);
}

// The Tdot epilogue of a batch: adds the G slabs [G][slab_rows][K] in group
// order, applies the intercept / centring correction and, per chain,
//   TD_OPER_UPD: q = d p + s g ; alpha = rho / (<p, d p> + <t, Omega t>) ;
//                x += alpha p ; r -= alpha q ; partials of r.r ; n_iter = k + 1
//   TD_RESID   : r = s (z + (phi eta2 - g)) - d x0 (warm) ; partials of r.r
//   TD_PLAIN   : r = g
// (spmv_csr.hip tdot_finalize_kernel per column).
template <int KC, int mode>
__global__ __launch_bounds__(VEC_BLOCK) void b_finalize_kernel(
    int64_t p, int K, int intercept, const double* __restrict__ slab, int G,
    int64_t slab_rows, const double* __restrict__ offset,
    const double* __restrict__ sumw_part, const double* __restrict__ s,
    const double* __restrict__ d, const double* __restrict__ pvec,
    double* __restrict__ x, double* __restrict__ r, ChainPtrs z, ChainPtrs phi,
    ChainPtrs eta2, int warm, double* __restrict__ rr_part,
    BatchState* __restrict__ bs, int cg_k, const double* __restrict__ pdp_part,
    const double* __restrict__ twt_part) {
  const int c0 = blockIdx.y * KC;
  const int64_t P = p + intercept;
  PartLoadK pw[KC], pa[KC], pb[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    pw[c] = part_issue_k(sumw_part + (c0 + c) * NPART);
    if (mode == TD_OPER_UPD) {
      pa[c] = part_issue_k(pdp_part + (c0 + c) * NPART);
      pb[c] = part_issue_k(twt_part + (c0 + c) * NPART);
    }
  }
  double sumw[KC], alpha[KC];
  bool run[KC];
  bool any = false;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    sumw[c] = part_finish_k(pw[c]);
    run[c] = true;
    alpha[c] = 0.;
    if (mode == TD_OPER_UPD) {
      run[c] = !bs->st[c0 + c].done;
      const double rho = bs->st[c0 + c].rho[cg_k & 1];
      const double pap = part_finish_k(pa[c]) + part_finish_k(pb[c]);
      alpha[c] = rho / pap;
    }
    any = any || run[c];
  }
  if (!any) return;
  double dacc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) dacc[c] = 0.;
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK) {
    double g[KC];
    if (intercept && jj == 0) {
#pragma unroll
      for (int c = 0; c < KC; ++c) g[c] = sumw[c];
    } else {
      const int64_t j = jj - intercept;
#pragma unroll
      for (int c = 0; c < KC; ++c) g[c] = 0.;
      for (int gi = 0; gi < G; ++gi) {
        const double* row = slab + ((int64_t)gi * slab_rows + j) * K + c0;
#pragma unroll
        for (int c = 0; c < KC; ++c) g[c] += row[c];
      }
      const double off = offset[j];
#pragma unroll
      for (int c = 0; c < KC; ++c) g[c] -= sumw[c] * off;
    }
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      if (!run[c]) continue;
      const int64_t at = jj * K + c0 + c;
      double rj;
      if (mode == TD_OPER_UPD) {
        const double pj = pvec[at];
        const double q = d[at] * pj + s[at] * g[c];
        x[at] += alpha[c] * pj;
        rj = r[at] - alpha[c] * q;
      } else if (mode == TD_RESID) {
        rj = s[at] * (z.p[c0 + c][jj] +
                      (phi.p[c0 + c][jj] * eta2.p[c0 + c][jj] - g[c]));
        if (warm) rj -= d[at] * x[at];
      } else {
        rj = g[c];  // TD_PLAIN: the product itself
      }
      r[at] = rj;
      dacc[c] += rj * rj;
    }
  }
  if (rr_part) block_store_partials_k<KC>(dacc, rr_part, c0);
  if (mode == TD_OPER_UPD && blockIdx.x == 0 && threadIdx.x < KC &&
      run[threadIdx.x])
    bs->st[c0 + threadIdx.x].n_iter = cg_k + 1;
}

// coef_c = s_c .* x_c   (cg_sampler.py:89)
template <int KC>
__global__ __launch_bounds__(VEC_BLOCK) void b_finish_kernel(
    int64_t P, int K, const double* __restrict__ s,
    const double* __restrict__ x, ChainOut coef) {
  const int c0 = blockIdx.y * KC;
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK) {
#pragma unroll
    for (int c = 0; c < KC; ++c)
      coef.p[c0 + c][jj] = s[jj * K + c0 + c] * x[jj * K + c0 + c];
  }
}

// grid of a batched vector kernel: NPART blocks x K / KC chain groups
#define BBX_KC_LAUNCH(K, KERNEL, STREAM, ...)                                  \
  do {                                                                         \
    if ((K) == 2)                                                              \
      BBX_LAUNCH((KERNEL<2>), dim3(NPART, 1), dim3(VEC_BLOCK), 0,      \
                         STREAM, __VA_ARGS__);                                 \
    else                                                                       \
      BBX_LAUNCH((KERNEL<4>), dim3(NPART, (K) / 4), dim3(VEC_BLOCK),   \
                         0, STREAM, __VA_ARGS__);                              \
  } while (0)
#define BBX_KC_LAUNCH_MODE(K, KERNEL, MODE, STREAM, ...)                       \
  do {                                                                         \
    if ((K) == 2)                                                              \
      BBX_LAUNCH((KERNEL<2, MODE>), dim3(NPART, 1), dim3(VEC_BLOCK),   \
                         0, STREAM, __VA_ARGS__);                              \
    else                                                                       \
      BBX_LAUNCH((KERNEL<4, MODE>), dim3(NPART, (K) / 4),              \
                         dim3(VEC_BLOCK), 0, STREAM, __VA_ARGS__);             \
  } while (0)

// ------------------------------------------------------------------- driver

static ChainPtrs gather_ptrs(const bbx_batch* b, DevMem bbx_chain::*field) {
  ChainPtrs out{};
  for (int c = 0; c < b->K; ++c) out.p[c] = (b->chain[c]->*field).as<double>();
  return out;
}

// Geometry of the operator as the batched vector kernels see it.  Sparse: the
// intercept entry and the implicit centring are corrections around the main
// block (sparse_matrix.py:68-129).  Dense: they are part of the stored matrix.
struct BatchShape {
  int icpt;            // 1: entry 0 of a P-vector is the intercept
  int64_t p_main;      // columns of the main block
  const double* offset;
};
static BatchShape batch_shape(const bbx_design* h) {
  BatchShape g;
  g.icpt = h->sparse ? h->intercept : 0;
  g.p_main = h->sparse ? h->p : h->P;
  g.offset = h->offset.as<double>();  // dense: P zeros
  return g;
}

// t_c = rowscale_c .* (X~ v_c) for the interleaved input, one pass over X.
static int batch_dot(bbx_batch* b, const double* v_il, const TiledBatchArgs& ba,
                     bool want_sums) {
  bbx_design* h = b->h;
  const int K = b->K;
  if (h->sparse)
    return launch_dot_tiled_k(h, K, v_il, bpart(b, PS_C), ba,
                              want_sums ? bpart(b, PS_SUMW) : nullptr,
                              want_sums ? (PS_TWT - PS_SUMW) * K * NPART : 0);
  return launch_dot_dense_k(h, K, v_il, ba,
                            want_sums ? bpart(b, PS_TWT) : nullptr);
}

// slabs of X~^T w_c for the interleaved input, one pass over X
static int batch_tdot(bbx_batch* b, const double* w_il, const double** slab,
                      int* G, int64_t* slab_rows) {
  bbx_design* h = b->h;
  if (h->sparse) {
    *slab_rows = h->p;
    return launch_tdot_tiled_k(h, b->K, w_il, slab, G);
  }
  *slab_rows = h->dense_ld;
  return launch_tdot_dense_k(h, b->K, w_il, slab, G);
}

// sum(w_c) partials of the Tdot epilogue: dense designs have no implicit
// centring, their epilogue reads zeros
static const double* batch_sumw(const bbx_batch* b) {
  return bpart(b, b->h->sparse ? PS_SUMW : PS_ZERO);
}

// psi_c = X~ coef_c for every chain in one pass (the linear predictor of the
// Omega update and of the log-likelihood, bayesbridge.py:401,407).
static int batch_linear_predictor(bbx_batch* b) {
  bbx_design* h = b->h;
  const int K = b->K;
  const BatchShape g = batch_shape(h);
  const ChainPtrs coef = gather_ptrs(b, &bbx_chain::coef);
  double* sp = b->sp.as<double>();
  BBX_KC_LAUNCH(K, b_prep_kernel, h->stream, h->P, b->KS, g.icpt, nullptr, coef,
                nullptr, g.offset, sp, bpart(b, PS_C));
  BBX_HIP(hipGetLastError());
  TiledBatchArgs ba;
  for (int c = 0; c < K; ++c) ba.out.p[c] = b->chain[c]->psi.as<double>();
  ba.out_stride = 1;
  ba.part_stride = NPART;
  return batch_dot(b, sp, ba, false);
}

// K draws of the prior-preconditioned CG sampler in lock step: cg_sampler.hip
// cg_sample_device per column, every product a batched one.
static int cg_sample_batch(bbx_batch* b, int maxiter, double atol,
                           bool cold, int* n_iter_out, int* info_out) {
  bbx_design* h = b->h;
  const int K = b->K, KS = b->KS;
  const int64_t P = h->P, n = h->n;
  const BatchShape g = batch_shape(h);
  hipStream_t st_ = h->stream;
  double* s = b->s.as<double>();
  double* d = b->d.as<double>();
  double* x = b->x.as<double>();
  double* r = b->r.as<double>();
  double* pvec = b->p.as<double>();
  double* sp = b->sp.as<double>();
  double* t = b->t.as<double>();
  double* w = b->w.as<double>();
  BatchState* bs = b->state.as<BatchState>();
  BatchState* host_bs = static_cast<BatchState*>(b->pinned);
  const ChainPtrs phi = gather_ptrs(b, &bbx_chain::phi);
  const ChainPtrs sd = gather_ptrs(b, &bbx_chain::sd);
  const ChainPtrs x0 = gather_ptrs(b, &bbx_chain::x0);
  const ChainPtrs z = gather_ptrs(b, &bbx_chain::z);
  const ChainPtrs omega = gather_ptrs(b, &bbx_chain::obs_prec);
  ChainPtrs e1{}, e2{};
  ChainOut coef{};
  const int n_unshrunk = b->chain[0]->n_unshrunk;
  for (int c = 0; c < K; ++c) {
    bbx_chain* ch = b->chain[c];
    const uint64_t seed = cg_draw_seed(ch, (uint64_t)ch->iter);
    BBX_TRY(launch_fill_normal(h, n, seed, STREAM_ETA1, b->eta1[c].as<double>()));
    BBX_TRY(launch_fill_normal(h, P, seed, STREAM_ETA2, b->eta2[c].as<double>()));
    e1.p[c] = b->eta1[c].as<double>();
    e2.p[c] = b->eta2[c].as<double>();
    coef.p[c] = ch->coef.as<double>();
  }
  TiledBatchArgs dot_args;
  for (int c = 0; c < K; ++c) {
    dot_args.rowscale.p[c] = omega.p[c];
    dot_args.out.p[c] = t + c;
  }
  dot_args.out_stride = KS;
  dot_args.part_stride = NPART;

  BBX_KC_LAUNCH(K, b_setup_kernel, st_, P, KS, n_unshrunk, phi, sd, x0, s, d, x,
                bs, atol);
  // r = b - A x0 through ONE transposed product (cg_sampler.hip, TD_RESID)
  const double* t0 = nullptr;
  if (!cold) {
    BBX_KC_LAUNCH(K, b_prep_kernel, st_, P, KS, g.icpt, x, ChainPtrs{}, s,
                  g.offset, sp, bpart(b, PS_C));
    BBX_HIP(hipGetLastError());
    BBX_TRY(batch_dot(b, sp, dot_args, false));
    t0 = t;
  }
  BBX_KC_LAUNCH(K, b_sqrt_scale_kernel, st_, n, KS, omega, e1, t0, w,
                bpart(b, PS_SUMW));
  BBX_HIP(hipGetLastError());
  const double* slab = nullptr;
  int G = 0;
  int64_t slab_rows = 0;
  BBX_TRY(batch_tdot(b, w, &slab, &G, &slab_rows));
  BBX_KC_LAUNCH_MODE(K, b_finalize_kernel, TD_RESID, st_, g.p_main, KS, g.icpt,
                     slab, G, slab_rows, g.offset, batch_sumw(b), s, d, pvec, x,
                     r, z, phi, e2, cold ? 0 : 1, bpart(b, PS_RR), bs, 0,
                     nullptr, nullptr);
  BBX_HIP(hipGetLastError());

  struct SkipScope {
    bbx_design* h;
    ~SkipScope() {
      h->skip_flag = nullptr;
      h->timer.cur_tag = -1;
    }
  } skip_scope{h};
  h->skip_flag = &bs->all_done;
  auto direction = [&](int kk) -> int {
    BBX_KC_LAUNCH(K, b_direction_kernel, st_, P, KS, K, g.icpt, kk, bs,
                  bpart(b, PS_RR), r, pvec, s, g.offset, sp, bpart(b, PS_C), d,
                  bpart(b, PS_PDP));
    BBX_HIP(hipGetLastError());
    return BBX_OK;
  };
  auto operator_and_update = [&](int kk) -> int {
    h->timer.cur_tag = kk;
    BBX_TRY(timer_begin(h, 2));
    BBX_TRY(batch_dot(b, sp, dot_args, true));
    BBX_TRY(batch_tdot(b, t, &slab, &G, &slab_rows));
    BBX_KC_LAUNCH_MODE(K, b_finalize_kernel, TD_OPER_UPD, st_, g.p_main, KS,
                       g.icpt, slab, G, slab_rows, g.offset, batch_sumw(b), s, d,
                       pvec, x, r, ChainPtrs{}, ChainPtrs{}, ChainPtrs{}, 0,
                       bpart(b, PS_RR), bs, kk, bpart(b, PS_PDP),
                       bpart(b, PS_TWT));
    BBX_HIP(hipGetLastError());
    return timer_end(h, 2);
  };
  auto finish = [&]() -> int {
    BBX_KC_LAUNCH(K, b_finish_kernel, st_, P, KS, s, x, coef);
    BBX_HIP(hipGetLastError());
    BBX_HIP(hipMemcpyAsync(host_bs, bs, sizeof(BatchState),
                           hipMemcpyDeviceToHost, st_));
    BBX_HIP(hipStreamSynchronize(st_));
    return BBX_OK;
  };
  // the schedule of cg_sample_device: run ahead to the previous solve's count
  // (the slowest chain's) + 2, look at the flags, then every other iteration
  int k = 0;
  bool finished_at_poll = false;
  int next_poll = b->last_cg_iter > 2 ? b->last_cg_iter + 2 : 1;
  for (;;) {
    const int stop = (next_poll < maxiter) ? next_poll : maxiter;
    for (; k < stop; ++k) {
      BBX_TRY(direction(k));
      BBX_TRY(operator_and_update(k));
    }
    if (k >= maxiter) break;
    BBX_TRY(direction(k));
    BBX_TRY(finish());
    if (host_bs->all_done) {
      finished_at_poll = true;
      break;
    }
    BBX_TRY(operator_and_update(k));
    ++k;
    next_poll = k + 2;
  }
  if (!finished_at_poll) BBX_TRY(finish());
  h->timer.cur_tag = -1;
  int slowest = 0;
  bool bad = false;
  for (int c = 0; c < K; ++c) {
    const CGState& cs = host_bs->st[c];
    slowest = std::max(slowest, cs.n_iter);
    n_iter_out[c] = cs.n_iter;
    info_out[c] = cs.bad ? -1 : (cs.done ? 0 : maxiter);
    bad = bad || cs.bad;
  }
  timer_drop_skipped(h, slowest);
  if (k > slowest) {  // launches past the last stop exited at entry
    h->n_dot -= (k - slowest);
    h->n_tdot -= (k - slowest);
  }
  b->last_cg_iter = slowest;
  if (bad) return fail(BBX_ERR_NUMERIC, "non-finite residual inside CG");
  return BBX_OK;
}

// One Gibbs iteration of every chain of the batch (bayesbridge.py:210-240).
static int batch_step(bbx_batch* b, int maxiter, double atol, int* n_cg,
                      int* info) {
  const int K = b->K;
  bool cold = true;
  for (int c = 0; c < K; ++c) {
    BBX_TRY(chain_pre_draw(b->chain[c]));
    cold = cold && b->chain[c]->mean_zero;
  }
  BBX_TRY(cg_sample_batch(b, maxiter, atol, cold, n_cg, info));
  for (int c = 0; c < K; ++c) b->chain[c]->mean_zero = false;
  BBX_TRY(batch_linear_predictor(b));
  // every chain's tau / lambda branch first (K second streams side by side:
  // the lambda kernel is latency bound), then the Omega updates one after the
  // other on the design's stream (they are ALU bound), then the joins
  // (at most BRANCH_STREAMS queues: chains c, c + 4, ... share one, in order)
  constexpr int BRANCH_STREAMS = 4;
  for (int phase : {POST_BRANCH, POST_MAIN, POST_JOIN})
    for (int c = 0; c < K; ++c)
      BBX_TRY(chain_post_draw(b->chain[c], true, phase,
                              b->chain[c % BRANCH_STREAMS]));
  return BBX_OK;
}

}  // namespace bbx

using namespace bbx;

extern "C" {

int bbx_batch_predict(bbx_design* design, int n_chain, double* speedup) {
  if (!design || !speedup) return fail(BBX_ERR_INVALID, "NULL argument");
  *speedup = 0.;
  if (!design_alive(design))
    return fail(BBX_ERR_INVALID, "not a live design handle");
  int slot = -1;
  for (int k = 0, w = 2; k < 5; ++k, w *= 2)
    if (n_chain == w) slot = k;
  if (slot < 0 || (design->sparse && n_chain > 4))
    return fail(BBX_ERR_INVALID, design->sparse
                                     ? "sparse designs batch 2 or 4 chains"
                                     : "a batch holds 2, 4, 8, 16 or 32 chains");
  // (asked per candidate width on every round of run_chains(batch='auto') and
  // again by bbx_batch_create: the answer of a design does not change)
  if (design->batch_speedup[slot] >= 0.) {
    *speedup = design->batch_speedup[slot];
    return BBX_OK;
  }
  BBX_HIP(hipSetDevice(design->device));
  if (design->sparse) {
    BBX_TRY(tiled_batch_predict(design, n_chain, speedup));
  } else {
    if (!dense_batch_applies(design))
      return fail(BBX_ERR_STATE,
                  "batched chains: this dense layout is not supported");
    BBX_TRY(dense_batch_predict(design, n_chain, speedup));
  }
  design->batch_speedup[slot] = *speedup;
  return BBX_OK;
}

int bbx_batch_create(bbx_design* design, int n_chain, bbx_chain* const* chains,
                     bbx_batch** out) {
  return bbx_batch_create_opts(design, n_chain, chains, 0u, out);
}

int bbx_batch_create_opts(bbx_design* design, int n_chain,
                          bbx_chain* const* chains, unsigned flags,
                          bbx_batch** out) {
  if (!out) return fail(BBX_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!design || !chains) return fail(BBX_ERR_INVALID, "NULL argument");
  if (n_chain != 2 && n_chain != 4 && n_chain != 8 && n_chain != 16 &&
      n_chain != 32)
    return fail(BBX_ERR_INVALID, "a batch holds 2, 4, 8, 16 or 32 chains");
  if (design->sparse && n_chain > 4)
    return fail(BBX_ERR_INVALID, "sparse designs batch 2 or 4 chains");
  for (int c = 0; c < n_chain; ++c) {
    if (!chains[c] || chains[c]->h != design)
      return fail(BBX_ERR_INVALID,
                  "every chain of a batch must be bound to the batch's design");
    for (int e = 0; e < c; ++e)
      if (chains[e] == chains[c])
        return fail(BBX_ERR_INVALID, "a chain appears twice in the batch");
    if (chains[c]->n_unshrunk != chains[0]->n_unshrunk)
      return fail(BBX_ERR_INVALID,
                  "the chains of a batch must agree on n_unshrunk");
  }
  bbx_design* h = design;
  if (h->sparse && h->format != BBX_FORMAT_TILED)
    return fail(BBX_ERR_STATE,
                "batched chains need the tiled format of a sparse design");
  if (!h->sparse && !dense_batch_applies(h))
    return fail(BBX_ERR_STATE,
                "batched chains: this dense layout is not supported");
  if (h->sparse && !tiled_batch_value_free(h) && n_chain > 2)
    return fail(BBX_ERR_INVALID,
                "designs with stored values batch at most 2 chains (four valued "
                "right-hand sides exceed the kernel's register budget)");
  BBX_HIP(hipSetDevice(h->device));
  if (!(flags & BBX_BATCH_ALLOW_SLOW)) {
    // a width the library's own cost model prices below single chains is an
    // error unless the caller asks for it (K = 4 at 1M x 50k ran at 0.975x)
    double pred = 0.;
    BBX_TRY(bbx_batch_predict(h, n_chain, &pred));
    if (pred < 1.) {
      char msg[256];
      snprintf(msg, sizeof(msg),
               "a batch of %d chains on this design is predicted to run at "
               "%.2fx the throughput of single chains (bbx_batch_predict); "
               "pass BBX_BATCH_ALLOW_SLOW to build it anyway", n_chain, pred);
      return fail(BBX_ERR_INVALID, msg);
    }
  }
  bbx_batch* b = new (std::nothrow) bbx_batch();
  if (!b) return fail(BBX_ERR_INVALID, "out of host memory");
  b->h = h;
  b->K = n_chain;
  for (int c = 0; c < n_chain; ++c) b->chain[c] = chains[c];
  b->KS = h->sparse ? n_chain : dense_batch_stride(n_chain);
  const size_t K = (size_t)n_chain, KS = (size_t)b->KS;
  // dense: the product kernels read whole 64-row / 64-column stages of their
  // 16-column operands; rows past P / n and columns past the chains stay zero
  const size_t rows_P = h->sparse ? (size_t)(h->P + 2)
                                  : (size_t)(h->dense_ld + DENSE_BATCH_ROW_PAD + 2);
  const size_t rows_n = h->sparse ? (size_t)h->n
                                  : (size_t)(h->n + DENSE_BATCH_ROW_PAD);
  auto body = [&]() -> int {
    if (h->sparse) BBX_TRY(ensure_tiled_k(h, n_chain));
    for (DevMem* m : {&b->s, &b->d, &b->x, &b->r, &b->p, &b->sp}) {
      BBX_TRY(m->alloc(sizeof(double) * rows_P * KS));
      BBX_HIP(hipMemset(m->ptr, 0, sizeof(double) * rows_P * KS));
    }
    for (DevMem* m : {&b->t, &b->w}) {
      BBX_TRY(m->alloc(sizeof(double) * rows_n * KS));
      BBX_HIP(hipMemset(m->ptr, 0, sizeof(double) * rows_n * KS));
    }
    for (int c = 0; c < n_chain; ++c) {
      BBX_TRY(b->eta1[c].alloc(sizeof(double) * (size_t)h->n));
      BBX_TRY(b->eta2[c].alloc(sizeof(double) * (size_t)h->P));
    }
    BBX_TRY(b->part.alloc(sizeof(double) * NPART * PS_COUNT * K));
    BBX_HIP(hipMemset(b->part.ptr, 0, sizeof(double) * NPART * PS_COUNT * K));
    BBX_TRY(b->state.alloc(sizeof(BatchState)));
    BBX_HIP(hipMemset(b->state.ptr, 0, sizeof(BatchState)));
    BBX_HIP(hipHostMalloc(&b->pinned, sizeof(BatchState) + 64,
                          hipHostMallocDefault));
    BBX_HIP(hipDeviceSynchronize());
    return BBX_OK;
  };
  const int st = no_throw(body);
  if (st < 0) {
    bbx_batch_destroy(b);
    return st;
  }
  *out = b;
  return BBX_OK;
}

int bbx_batch_destroy(bbx_batch* b) {
  if (!b) return BBX_OK;
  if (b->h && design_alive(b->h)) {   // (see bbx_chain_destroy)
    (void)hipSetDevice(b->h->device);
    (void)hipStreamSynchronize(b->h->stream);
  }
  if (b->pinned) (void)hipHostFree(b->pinned);
  delete b;  // the chains and the design stay the caller's
  return BBX_OK;
}

static int bbx_batch_run_impl(bbx_batch* b, int n_iter, int n_burnin, int thin,
                              int maxiter, double atol,
                              double* const* d_coef, double* gscale,
                              double* logp, double* n_cg_iter) {
  if (!b || !b->h) return fail(BBX_ERR_INVALID, "batch handle is NULL");
  if (n_iter < 0 || n_burnin < 0 || thin < 1 || n_burnin > n_iter)
    return fail(BBX_ERR_INVALID, "bad n_iter / n_burnin / thin");
  bbx_design* h = b->h;
  BBX_HIP(hipSetDevice(h->device));
  if (maxiter <= 0) maxiter = 500;                      // reg_coef_sampler.py:95
  if (!(atol > 0.)) atol = 10e-6 * std::sqrt((double)h->P);
  const int K = b->K;
  const int n_sample = (n_iter - n_burnin) / thin;
  for (int c = 0; c < K; ++c) BBX_TRY(chain_begin_run(b->chain[c], n_sample));
  int n_unconverged = 0;
  for (int c = 0; c < K; ++c) b->n_unconverged[c] = 0;
  for (int it = 1; it <= n_iter; ++it) {
    int ncg[BATCH_MAX] = {}, info[BATCH_MAX] = {};
    BBX_TRY(batch_step(b, maxiter, atol, ncg, info));
    for (int c = 0; c < K; ++c)
      if (info[c] > 0) {
        ++n_unconverged;
        ++b->n_unconverged[c];
      }
    if (it <= n_burnin || (it - n_burnin) % thin != 0) continue;
    const int idx = (it - n_burnin) / thin - 1;  // gibbs_util.py:170
    if (idx >= n_sample) continue;
    for (int c = 0; c < K; ++c) {
      BBX_TRY(chain_save_sample(b->chain[c], idx,
                                d_coef ? d_coef[c] : nullptr, nullptr,
                                nullptr));
      if (n_cg_iter) n_cg_iter[(size_t)c * n_sample + idx] = (double)ncg[c];
    }
  }
  for (int c = 0; c < K; ++c)
    BBX_TRY(chain_end_run(b->chain[c], n_sample,
                          gscale ? gscale + (size_t)c * n_sample : nullptr,
                          logp ? logp + (size_t)c * n_sample : nullptr));
  return n_unconverged;
}

int bbx_batch_run(bbx_batch* b, int n_iter, int n_burnin, int thin,
                  int maxiter, double atol, double* const* d_coef,
                  double* gscale, double* logp, double* n_cg_iter) {
  return no_throw([&]() -> int {
    return bbx_batch_run_impl(b, n_iter, n_burnin, thin, maxiter, atol, d_coef,
                              gscale, logp, n_cg_iter);
  });
}

// The batched products on their own (host pointers, chain-major [K][len]):
// out_c = X~ v_c and out_c = X~^T w_c through the kernels the batch's CG loop
// uses.  Exposed for the parity tests (kernel == CPU emulator per column).
int bbx_batch_dot(bbx_batch* b, const double* v, double* out) {
  return no_throw([&]() -> int {
    if (!b || !b->h || !v || !out) return fail(BBX_ERR_INVALID, "NULL argument");
    bbx_design* h = b->h;
    const int K = b->K;
    const int64_t P = h->P, n = h->n;
    const BatchShape g = batch_shape(h);
    BBX_HIP(hipSetDevice(h->device));
    DevMem dv, dout;
    BBX_TRY(dv.alloc(sizeof(double) * (size_t)(K * P)));
    BBX_TRY(dout.alloc(sizeof(double) * (size_t)(K * n)));
    BBX_HIP(hipMemcpyAsync(dv.ptr, v, sizeof(double) * (size_t)(K * P),
                           hipMemcpyHostToDevice, h->stream));
    ChainPtrs src{};
    TiledBatchArgs ba;
    for (int c = 0; c < K; ++c) {
      src.p[c] = dv.as<double>() + (size_t)c * P;
      ba.out.p[c] = dout.as<double>() + (size_t)c * n;
    }
    ba.out_stride = 1;
    ba.part_stride = NPART;
    double* sp = b->sp.as<double>();
    BBX_KC_LAUNCH(K, b_prep_kernel, h->stream, P, b->KS, g.icpt, nullptr, src,
                  nullptr, g.offset, sp, bpart(b, PS_C));
    BBX_HIP(hipGetLastError());
    BBX_TRY(batch_dot(b, sp, ba, false));
    BBX_HIP(hipMemcpyAsync(out, dout.ptr, sizeof(double) * (size_t)(K * n),
                           hipMemcpyDeviceToHost, h->stream));
    BBX_HIP(hipStreamSynchronize(h->stream));
    return BBX_OK;
  });
}

int bbx_batch_tdot(bbx_batch* b, const double* w, double* out) {
  return no_throw([&]() -> int {
    if (!b || !b->h || !w || !out) return fail(BBX_ERR_INVALID, "NULL argument");
    bbx_design* h = b->h;
    const int K = b->K;
    const int64_t P = h->P, n = h->n;
    const BatchShape g = batch_shape(h);
    BBX_HIP(hipSetDevice(h->device));
    // interleave on the host: the kernels take [n][K]
    const int KS = b->KS;
    std::vector<double> il((size_t)(KS * n), 0.), res((size_t)(KS * P));
    for (int c = 0; c < K; ++c)
      for (int64_t i = 0; i < n; ++i) il[(size_t)(i * KS + c)] = w[(size_t)c * n + i];
    double* dw = b->w.as<double>();
    BBX_HIP(hipMemcpyAsync(dw, il.data(), sizeof(double) * il.size(),
                           hipMemcpyHostToDevice, h->stream));
    // partials of sum(w_c) through the scaling kernel: w = w - sqrt(0) * 0
    DevMem zeros;
    BBX_TRY(zeros.alloc(sizeof(double) * (size_t)n));
    BBX_HIP(hipMemsetAsync(zeros.ptr, 0, sizeof(double) * (size_t)n, h->stream));
    ChainPtrs zp{};
    for (int c = 0; c < K; ++c) zp.p[c] = zeros.as<double>();
    BBX_KC_LAUNCH(K, b_sqrt_scale_kernel, h->stream, n, KS, zp, zp, dw, dw,
                  bpart(b, PS_SUMW));
    BBX_HIP(hipGetLastError());
    const double* slab = nullptr;
    int G = 0;
    int64_t slab_rows = 0;
    BBX_TRY(batch_tdot(b, dw, &slab, &G, &slab_rows));
    double* r = b->r.as<double>();
    BBX_KC_LAUNCH_MODE(K, b_finalize_kernel, TD_PLAIN, h->stream, g.p_main, KS,
                       g.icpt, slab, G, slab_rows, g.offset, batch_sumw(b),
                       nullptr, nullptr, nullptr, nullptr, r, ChainPtrs{},
                       ChainPtrs{}, ChainPtrs{}, 0, nullptr,
                       b->state.as<BatchState>(), 0, nullptr, nullptr);
    BBX_HIP(hipGetLastError());
    BBX_HIP(hipMemcpyAsync(res.data(), r, sizeof(double) * res.size(),
                           hipMemcpyDeviceToHost, h->stream));
    BBX_HIP(hipStreamSynchronize(h->stream));
    for (int c = 0; c < K; ++c)
      for (int64_t j = 0; j < P; ++j) out[(size_t)c * P + j] = res[(size_t)(j * KS + c)];
    return BBX_OK;
  });
}

// Same with HOST coefficient buffers [n_chain][n_sample][P] (copied at the end).
int bbx_batch_run_host(bbx_batch* b, int n_iter, int n_burnin, int thin,
                       int maxiter, double atol, double* coef, double* gscale,
                       double* logp, double* n_cg_iter) {
  return no_throw([&]() -> int {
    if (!b || !b->h) return fail(BBX_ERR_INVALID, "batch handle is NULL");
    if (n_iter < 0 || n_burnin < 0 || thin < 1 || n_burnin > n_iter)
      return fail(BBX_ERR_INVALID, "bad n_iter / n_burnin / thin");
    BBX_HIP(hipSetDevice(b->h->device));
    const int64_t P = b->h->P;
    const int64_t n_sample = (n_iter - n_burnin) / thin;
    DevMem dc[BATCH_MAX];
    double* ptrs[BATCH_MAX] = {};
    if (coef)
      for (int c = 0; c < b->K; ++c) {
        BBX_TRY(dc[c].alloc(sizeof(double) * (size_t)(n_sample * P + 1)));
        ptrs[c] = dc[c].as<double>();
      }
    const int st = bbx_batch_run_impl(b, n_iter, n_burnin, thin, maxiter, atol,
                                      coef ? ptrs : nullptr, gscale, logp,
                                      n_cg_iter);
    if (st < 0) return st;
    if (coef && n_sample > 0)
      for (int c = 0; c < b->K; ++c)
        BBX_HIP(hipMemcpy(coef + (size_t)c * n_sample * P, dc[c].ptr,
                          sizeof(double) * (size_t)(n_sample * P),
                          hipMemcpyDeviceToHost));
    return st;
  });
}

int bbx_batch_unconverged(const bbx_batch* b, int* per_chain) {
  if (!b || !per_chain) return fail(BBX_ERR_INVALID, "NULL argument");
  for (int c = 0; c < b->K; ++c) per_chain[c] = b->n_unconverged[c];
  return BBX_OK;
}

int bbx_batch_bytes(const bbx_batch* b, int64_t* dot_bytes,
                    int64_t* tdot_bytes) {
  if (!b || !b->h) return fail(BBX_ERR_INVALID, "batch handle is NULL");
  int64_t db = 0, tb = 0;
  if (b->h->sparse) BBX_TRY(tiled_batch_bytes(b->h, b->K, &db, &tb));
  else BBX_TRY(dense_batch_bytes(b->h, b->K, &db, &tb));
  if (dot_bytes) *dot_bytes = db;
  if (tdot_bytes) *tdot_bytes = tb;
  return BBX_OK;
}

}  // extern "C"
