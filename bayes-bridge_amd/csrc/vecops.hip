// P- and n-length vector kernels around the operator: input scaling, the
// partial sums that feed the intercept/centring corrections, and the fused
// CG recurrences.  Every reduction is two-stage with NPART partials that the
// consumer re-adds in a fixed order (bitwise reproducible).
#include "common.hpp"

namespace bbx {

__device__ inline double wave_sum_v(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, WAVE);
  return x;
}

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

// v = s .* x (or x), c_part = partials of <offset, v[1:]>.
__global__ __launch_bounds__(VEC_BLOCK) void prep_v_kernel(
    int64_t P, int intercept, const double* __restrict__ x,
    const double* __restrict__ s, const double* __restrict__ offset,
    double* __restrict__ v, double* __restrict__ c_part) {
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

__global__ __launch_bounds__(VEC_BLOCK) void sqrt_scale_kernel(
    int64_t len, const double* __restrict__ omega,
    const double* __restrict__ eta, double* __restrict__ w,
    double* __restrict__ part) {
  double acc = 0.;
  for (int64_t i = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * VEC_BLOCK) {
    const double val = sqrt(omega[i]) * eta[i];
    w[i] = val;
    acc += val;
  }
  block_store_partial(acc, part);
}

int launch_prep_v(bbx_design* h, const double* d_x, const double* d_s,
                  double* d_v, double* d_c_part) {
  hipLaunchKernelGGL(prep_v_kernel, dim3(NPART), dim3(VEC_BLOCK), 0, h->stream,
                     h->P, h->intercept, d_x, d_s, h->offset.as<double>(), d_v,
                     d_c_part);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_sum_n(bbx_design* h, const double* d_w, int64_t len,
                 double* d_part) {
  hipLaunchKernelGGL(sum_n_kernel, dim3(NPART), dim3(VEC_BLOCK), 0, h->stream,
                     len, d_w, d_part);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_sqrt_scale(bbx_design* h, const double* d_omega,
                      const double* d_eta, double* d_w, double* d_part) {
  hipLaunchKernelGGL(sqrt_scale_kernel, dim3(NPART), dim3(VEC_BLOCK), 0,
                     h->stream, h->n, d_omega, d_eta, d_w, d_part);
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
    double* __restrict__ s, double* __restrict__ d, double* __restrict__ xs) {
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

// r = b - q, partials of r.r   (SciPy cg: r = b - A x0)
__global__ __launch_bounds__(VEC_BLOCK) void cg_init_resid_kernel(
    int64_t P, const double* __restrict__ b, const double* __restrict__ q,
    double* __restrict__ r, double* __restrict__ rr_part) {
  double acc = 0.;
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK) {
    const double val = b[jj] - q[jj];
    r[jj] = val;
    acc += val * val;
  }
  block_store_partial(acc, rr_part);
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
    double* __restrict__ c_part) {
  if (st->done) return;
  const double rho = sum_partials_v(rr_part);
  const bool finite = (rho == rho) && (rho - rho == 0.);
  if (!finite || sqrt(rho) < st->atol) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      st->done = 1;
      if (!finite) st->bad = 1;
    }
    return;
  }
  const double beta = (k > 0) ? rho / st->rho[(k - 1) & 1] : 0.;
  double acc = 0.;
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK) {
    double pj = r[jj];
    if (k > 0) pj += beta * pvec[jj];
    pvec[jj] = pj;
    const double v = s[jj] * pj;
    sp[jj] = v;
    if (jj >= intercept) acc += offset[jj - intercept] * v;
  }
  block_store_partial(acc, c_part);
  if (blockIdx.x == 0 && threadIdx.x == 0) st->rho[k & 1] = rho;
}

// alpha = rho / (p.q); x += alpha p; r -= alpha q; partials of the new r.r.
__global__ __launch_bounds__(VEC_BLOCK) void cg_update_kernel(
    int64_t P, int k, CGState* __restrict__ st,
    const double* __restrict__ pq_part, const double* __restrict__ pvec,
    const double* __restrict__ q, double* __restrict__ x,
    double* __restrict__ r, double* __restrict__ rr_part) {
  if (st->done) return;
  const double pq = sum_partials_v(pq_part);
  const double alpha = st->rho[k & 1] / pq;
  double acc = 0.;
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK) {
    x[jj] += alpha * pvec[jj];
    const double rj = r[jj] - alpha * q[jj];
    r[jj] = rj;
    acc += rj * rj;
  }
  block_store_partial(acc, rr_part);
  if (blockIdx.x == 0 && threadIdx.x == 0) st->n_iter = k + 1;
}

// coef = s .* x   (cg_sampler.py:89)
__global__ __launch_bounds__(VEC_BLOCK) void cg_finish_kernel(
    int64_t P, const double* __restrict__ s, const double* __restrict__ x,
    double* __restrict__ coef) {
  for (int64_t jj = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x; jj < P;
       jj += (int64_t)gridDim.x * VEC_BLOCK)
    coef[jj] = s[jj] * x[jj];
}

int launch_cg_setup(bbx_design* h, int n_unshrunk, const double* phi,
                    const double* sd, const double* x0, double* s, double* d,
                    double* xs) {
  hipLaunchKernelGGL(cg_setup_kernel, dim3(NPART), dim3(VEC_BLOCK), 0,
                     h->stream, h->P, n_unshrunk, phi, sd, x0, s, d, xs);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_cg_init_resid(bbx_design* h, const double* b, const double* q,
                         double* r, double* rr_part) {
  hipLaunchKernelGGL(cg_init_resid_kernel, dim3(NPART), dim3(VEC_BLOCK), 0,
                     h->stream, h->P, b, q, r, rr_part);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_cg_direction(bbx_design* h, int k, CGState* st,
                        const double* rr_part, const double* r, double* pvec,
                        const double* s, double* sp, double* c_part) {
  hipLaunchKernelGGL(cg_direction_kernel, dim3(NPART), dim3(VEC_BLOCK), 0,
                     h->stream, h->P, h->intercept, k, st, rr_part, r, pvec, s,
                     h->offset.as<double>(), sp, c_part);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_cg_update(bbx_design* h, int k, CGState* st, const double* pq_part,
                     const double* pvec, const double* q, double* x, double* r,
                     double* rr_part) {
  hipLaunchKernelGGL(cg_update_kernel, dim3(NPART), dim3(VEC_BLOCK), 0,
                     h->stream, h->P, k, st, pq_part, pvec, q, x, r, rr_part);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_cg_finish(bbx_design* h, const double* s, const double* x,
                     double* coef) {
  hipLaunchKernelGGL(cg_finish_kernel, dim3(NPART), dim3(VEC_BLOCK), 0,
                     h->stream, h->P, s, x, coef);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

}  // namespace bbx
