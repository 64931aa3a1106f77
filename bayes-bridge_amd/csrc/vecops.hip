// P- and n-length vector kernels around the operator: input scaling, the
// partial sums that feed the intercept/centring corrections, and the fused
// CG recurrences.  Every reduction is two-stage with NPART partials that the
// consumer re-adds in a fixed order (bitwise reproducible).
#include "common.hpp"

namespace bbx {

std::atomic<unsigned long long> g_launch_count{0};


__device__ inline double wave_sum_v(double x) { return wave_allsum(x); }

__device__ inline void block_store_partial(double x, double* part) {
  __shared__ double s_w[VEC_BLOCK / WAVE];
  x = wave_sum_v(x);
  if ((threadIdx.x & (WAVE - 1)) == 0) s_w[threadIdx.x / WAVE] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = 0.;
#pragma unroll
    for (int k = 0; k < VEC_BLOCK / WAVE; ++k) r += s_w[k];
    part[blockIdx.x] = r;
  }
  __syncthreads();
}

__device__ inline double sum_partials_v(const double* part) {
  __shared__ double s_tot;
  if (threadIdx.x < WAVE) {
    double a = 0.;
#pragma unroll
    for (int k = 0; k < NPART / WAVE; ++k) a += part[threadIdx.x + k * WAVE];
    a = wave_sum_v(a);
    if (threadIdx.x == 0) s_tot = a;
  }
  __syncthreads();
  const double r = s_tot;
  __syncthreads();
  return r;
}

// Split-phase form of sum_partials_v: the loads are ISSUED first so that the
// caller can queue its own independent loads (vector elements, CG scalars)
// behind them, and everything comes back in ONE memory round trip instead of
// one per dependent step.  The P-vector kernels are pure latency (~5 us for
// 400 KB), so the number of serial round trips is their cost.  Same adds in
// the same order as sum_partials_v: bit-identical result.
struct PartLoad {
  double v[NPART / WAVE];
};
__device__ inline PartLoad part_issue(const double* part) {
  PartLoad p;
#pragma unroll
  for (int k = 0; k < NPART / WAVE; ++k)
    p.v[k] = part[(threadIdx.x & (WAVE - 1)) + k * WAVE];  // branch-free
  return p;
}
__device__ inline double part_finish(const PartLoad& p) {
  __shared__ double s_tot2;
  if (threadIdx.x < WAVE) {
    double a = 0.;
#pragma unroll
    for (int k = 0; k < NPART / WAVE; ++k) a += p.v[k];
    a = wave_sum_v(a);
    if (threadIdx.x == 0) s_tot2 = a;
  }
  __syncthreads();
  const double r = s_tot2;
  __syncthreads();
  return r;
}

// v = s .* x (or x), c_part = partials of <offset, v[1:]>.
__global__ __launch_bounds__(VEC_BLOCK) void prep_v_kernel(
    int64_t P, int intercept, const double* __restrict__ x,
    const double* __restrict__ s, const double* __restrict__ offset,
    double* __restrict__ v, double* __restrict__ c_part,
    const int* __restrict__ skip) {
  if (skip && *skip) return;   // (work enqueued behind a CG stop test)
  double acc = 0.;
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK) {
    double val = x[jj];
    if (s) val *= s[jj];
    if (v) v[jj] = val;
    if (jj >= intercept) acc += offset[jj - intercept] * val;
  }
  block_store_partial(acc, c_part);
}

__global__ __launch_bounds__(VEC_BLOCK) void sum_n_kernel(
    int64_t len, const double* __restrict__ w, double* __restrict__ part) {
  double acc = 0.;
  for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * VEC_BLOCK)
    acc += w[i];
  block_store_partial(acc, part);
}

// w = sqrt(omega) eta        (plain; cg_sampler.py:66)
// w = -sqrt(omega) eta       (negate)
// w = minus - sqrt(omega) eta (minus != nullptr), and the partials of sum(w).
__global__ __launch_bounds__(VEC_BLOCK) void sqrt_scale_kernel(
    int64_t len, const double* __restrict__ omega,
    const double* __restrict__ eta, double* __restrict__ w,
    double* __restrict__ part, const double* __restrict__ minus, int negate) {
  double acc = 0.;
  for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * VEC_BLOCK) {
    double val = sqrt(omega[i]) * eta[i];
    if (minus) val = minus[i] - val;
    else if (negate) val = -val;
    w[i] = val;
    acc += val;
  }
  block_store_partial(acc, part);
}

int launch_prep_v(bbx_design* h, const double* d_x, const double* d_s,
                  double* d_v, double* d_c_part, const int* d_skip) {
  BBX_LAUNCH(prep_v_kernel, dim3(NPART), dim3(VEC_BLOCK), 0, h->stream,
                     h->P, h->intercept, d_x, d_s, h->offset.as<double>(), d_v,
                     d_c_part, d_skip);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_sum_n(bbx_design* h, const double* d_w, int64_t len,
                 double* d_part) {
  BBX_LAUNCH(sum_n_kernel, dim3(NPART), dim3(VEC_BLOCK), 0, h->stream,
                     len, d_w, d_part);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_sqrt_scale(bbx_design* h, const double* d_omega,
                      const double* d_eta, double* d_w, double* d_part,
                      const double* d_minus, bool negate) {
  BBX_LAUNCH(sqrt_scale_kernel, dim3(NPART), dim3(VEC_BLOCK), 0,
                     h->stream, h->n, d_omega, d_eta, d_w, d_part, d_minus,
                     negate ? 1 : 0);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

// ---------------------------------------------------------------- CG kernels

// s, d and the scaled warm start (cg_sampler.py:104,128-138,76):
//   s_j = 1/phi_j (j >= n_unshrunk),  2*sd_j (j < n_unshrunk)
//   d_j = (s_j phi_j)^2 ;  xs_j = x0_j / s_j
__global__ __launch_bounds__(VEC_BLOCK) void cg_setup_kernel(
    int64_t P, int n_unshrunk, const double* __restrict__ phi,
    const double* __restrict__ sd, const double* __restrict__ x0,
    double* __restrict__ s, double* __restrict__ d, double* __restrict__ xs,
    CGState* __restrict__ st, double atol) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // the solve's scalars start here, on the device: nothing to upload
    st->rho[0] = st->rho[1] = 0.;
    st->atol = atol;
    st->bnorm2 = 0.;
    st->n_iter = 0;
    st->done = 0;
    st->bad = 0;
    st->pad = 0;
    st->running = 1;
    st->pad2 = 0;
  }
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK) {
    const double ph = phi[jj];
    const double sj = (jj < n_unshrunk) ? 2. * sd[jj] : 1. / ph;
    const double sp = sj * ph;
    s[jj] = sj;
    d[jj] = sp * sp;
    xs[jj] = x0[jj] / sj;
  }
}

// Top of CG iteration k: stop test, rho, search direction, scaled copy for the
// operator and the partials of <offset, (s.*p)[1:]>.
//   if ||r|| < atol: done                      (SciPy _isolve cg loop top)
//   rho = r.r ; p = r + (rho/rho_prev) p  (k > 0)  |  p = r  (k == 0)
__global__ __launch_bounds__(VEC_BLOCK) void cg_direction_kernel(
    int64_t P, int intercept, int k, CGState* __restrict__ st,
    const double* __restrict__ rr_part, const double* __restrict__ r,
    double* __restrict__ pvec, const double* __restrict__ s,
    const double* __restrict__ offset, double* __restrict__ sp,
    double* __restrict__ c_part, const double* __restrict__ d,
    double* __restrict__ pdp_part, unsigned long long* word,
    unsigned long long tag) {
  // every load that does not depend on another one goes out first
  const PartLoad pl = part_issue(rr_part);
  const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
  const int64_t j0 = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x;
  const bool has0 = j0 < P;
  double r0 = 0., p0 = 0., s0 = 0., o0 = 0., d0 = 0.;
  if (has0) {
    r0 = r[j0];
    if (k > 0) p0 = pvec[j0];
    s0 = s[j0];
    if (j0 >= intercept) o0 = offset[j0 - intercept];
    if (pdp_part) d0 = d[j0];
  }
  const int was_done = st->done;
  const double atol = st->atol;
  const double rho_prev = (k > 0) ? st->rho[(k - 1) & 1] : 1.;
  if (was_done) return;
  const double rho = part_finish(pl);
  const bool finite = (rho == rho) && (rho - rho == 0.);
  if (!finite || sqrt(rho) < atol) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      st->done = 1;
      st->running = 0;
      if (!finite) st->bad = 1;
      // the host learns it from here: iteration count = k (SciPy's callbacks)
      if (word)
        cg_word_store(word, tag | CG_WORD_DONE | (finite ? 0ull : CG_WORD_BAD) |
                                (unsigned long long)k);
    }
    return;
  }
  // stop test k passed: the host may enqueue the next iteration(s)
  if (word && blockIdx.x == 0 && threadIdx.x == 0)
    cg_word_store(word, tag | (unsigned long long)(k + 1));
  const double beta = (k > 0) ? rho / rho_prev : 0.;
  double acc = 0., acc_d = 0.;
  if (has0) {
    double pj = r0;
    if (k > 0) pj += beta * p0;
    pvec[j0] = pj;
    const double v = s0 * pj;
    sp[j0] = v;
    if (j0 >= intercept) acc += o0 * v;
    acc_d += d0 * pj * pj;
  }
  for (int64_t jj = j0 + stride; jj < P; jj += stride) {
    double pj = r[jj];
    if (k > 0) pj += beta * pvec[jj];
    pvec[jj] = pj;
    const double v = s[jj] * pj;
    sp[jj] = v;
    if (jj >= intercept) acc += offset[jj - intercept] * v;
    if (pdp_part) acc_d += d[jj] * pj * pj;
  }
  block_store_partial(acc, c_part);
  // <p, d p>: the diagonal half of the curvature p.Ap (TD_OPER_UPD epilogue)
  if (pdp_part) block_store_partial(acc_d, pdp_part);
  if (blockIdx.x == 0 && threadIdx.x == 0) st->rho[k & 1] = rho;
}

// alpha = rho / (p.q); x += alpha p; r -= alpha q; partials of the new r.r.
__global__ __launch_bounds__(VEC_BLOCK) void cg_update_kernel(
    int64_t P, int k, CGState* __restrict__ st,
    const double* __restrict__ pq_part, const double* __restrict__ pvec,
    const double* __restrict__ q, double* __restrict__ x,
    double* __restrict__ r, double* __restrict__ rr_part) {
  const PartLoad pl = part_issue(pq_part);
  const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
  const int64_t j0 = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x;
  const bool has0 = j0 < P;
  double p0 = 0., q0 = 0., x0 = 0., r0 = 0.;
  if (has0) {
    p0 = pvec[j0];
    q0 = q[j0];
    x0 = x[j0];
    r0 = r[j0];
  }
  const int was_done = st->done;
  const double rho = st->rho[k & 1];
  if (was_done) return;
  const double pq = part_finish(pl);
  const double alpha = rho / pq;
  double acc = 0.;
  if (has0) {
    x[j0] = x0 + alpha * p0;
    const double rj = r0 - alpha * q0;
    r[j0] = rj;
    acc += rj * rj;
  }
  for (int64_t jj = j0 + stride; jj < P; jj += stride) {
    x[jj] += alpha * pvec[jj];
    const double rj = r[jj] - alpha * q[jj];
    r[jj] = rj;
    acc += rj * rj;
  }
  block_store_partial(acc, rr_part);
  if (blockIdx.x == 0 && threadIdx.x == 0) st->n_iter = k + 1;
}

// coef = s .* x   (cg_sampler.py:89)
// `copy`: a second destination (bbx_design::coef_copy: the sample slot of a
// chain's kept iteration, instead of a D2D copy after the solve).
__global__ __launch_bounds__(VEC_BLOCK) void cg_finish_kernel(
    int64_t P, const double* __restrict__ s, const double* __restrict__ x,
    double* __restrict__ coef, double* __restrict__ copy) {
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK) {
    const double v = s[jj] * x[jj];
    coef[jj] = v;
    if (copy) copy[jj] = v;
  }
}

int launch_cg_setup(bbx_design* h, int n_unshrunk, const double* phi,
                    const double* sd, const double* x0, double* s, double* d,
                    double* xs, CGState* st, double atol) {
  BBX_LAUNCH(cg_setup_kernel, dim3(NPART), dim3(VEC_BLOCK), 0,
                     h->stream, h->P, n_unshrunk, phi, sd, x0, s, d, xs, st,
                     atol);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_cg_direction(bbx_design* h, int k, CGState* st,
                        const double* rr_part, const double* r, double* pvec,
                        const double* s, double* sp, double* c_part,
                        const double* d, double* pdp_part,
                        unsigned long long* word, unsigned long long tag) {
  BBX_LAUNCH(cg_direction_kernel, dim3(NPART), dim3(VEC_BLOCK), 0,
                     h->stream, h->P, h->intercept, k, st, rr_part, r, pvec, s,
                     h->offset.as<double>(), sp, c_part, d, pdp_part, word, tag);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_cg_update(bbx_design* h, int k, CGState* st, const double* pq_part,
                     const double* pvec, const double* q, double* x, double* r,
                     double* rr_part) {
  BBX_LAUNCH(cg_update_kernel, dim3(NPART), dim3(VEC_BLOCK), 0,
                     h->stream, h->P, k, st, pq_part, pvec, q, x, r, rr_part);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_cg_finish(bbx_design* h, const double* s, const double* x,
                     double* coef) {
  BBX_LAUNCH(cg_finish_kernel, dim3(NPART), dim3(VEC_BLOCK), 0,
                     h->stream, h->P, s, x, coef, h->coef_copy);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

}  // namespace bbx
