// Reference-layout sparse operator (BBX_FORMAT_CSR): f64 values + i32 indices,
// CSR of X for  X v  and CSR of X^T for  X^T w.
//
// Replaces SparseDesignMatrix.main_dot / main_Tdot
// (design_matrix/sparse_matrix.py:87-101,119-129), whose arithmetic is SciPy's
// csr_matvec / csc_matvec, with the intercept and centring corrections of
// sparse_matrix.py:77-81,108-114,127-128 fused in:
//     dot :  t_i = rowscale_i * (c + sum_k X_ik v_{1+k}),  c = v_0 - <offset, v_1:>
//     Tdot:  g   = [sum w ; X^T w - sum(w) offset], then the CG epilogues.
//
// Both kernels are HBM-bound streams of (value, index) pairs with a gather
// from an L2-resident vector; see DESIGN.md for the roofline accounting.
#include <hipcub/hipcub.hpp>

#include "common.hpp"

namespace bbx {

constexpr int T_CHUNK = 2048;  // stored entries of X^T handled by one wave

__device__ inline double wave_sum(double x) { return wave_allsum(x); }

// Sum of the NPART partials, identical in every block (fixed order).
__device__ inline double sum_partials(const double* part) {
  __shared__ double s_tot;
  if (threadIdx.x < WAVE) {
    double a = 0.;
#pragma unroll
    for (int k = 0; k < NPART / WAVE; ++k) a += part[threadIdx.x + k * WAVE];
    a = wave_sum(a);
    if (threadIdx.x == 0) s_tot = a;
  }
  __syncthreads();
  double r = s_tot;
  __syncthreads();
  return r;
}

// G lanes cooperate on one row; 256/G rows per block pass.
template <int G, bool BIN>
__global__ __launch_bounds__(256) void csr_dot_kernel(
    int64_t n, const int32_t* __restrict__ indptr,
    const int32_t* __restrict__ indices, const double* __restrict__ data,
    const double* __restrict__ v, int intercept,
    const double* __restrict__ c_part, const double* __restrict__ rowscale,
    double* __restrict__ out, const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;  // the CG solve has already stopped
  const double csum = sum_partials(c_part);
  const double c = (intercept ? v[0] : 0.) - csum;
  const double* __restrict__ vm = v + intercept;
  const int sub = threadIdx.x % G;
  const int64_t rows_per_block = 256 / G;
  for (int64_t row = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / G;
       row < n; row += (int64_t)gridDim.x * rows_per_block) {
    const int32_t b = indptr[row], e = indptr[row + 1];
    double acc = 0.;
    for (int32_t k = b + sub; k < e; k += G) {
      const double x = vm[indices[k]];
      acc += BIN ? x : data[k] * x;
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, G);
    if (sub == 0) {
      double t = c + acc;
      if (rowscale) t *= rowscale[row];
      out[row] = t;
    }
  }
}

// One wave per chunk of a row of X^T.
template <bool BIN>
__global__ __launch_bounds__(256) void csr_tdot_chunk_kernel(
    int64_t n_chunk, const int32_t* __restrict__ chunk_row,
    const int32_t* __restrict__ chunk_begin,
    const int32_t* __restrict__ t_indptr,
    const int32_t* __restrict__ t_indices, const double* __restrict__ t_data,
    const double* __restrict__ w, double* __restrict__ partial,
    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;  // the CG solve has already stopped
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t c = (int64_t)blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  if (c >= n_chunk) return;
  const int32_t row = chunk_row[c];
  const int32_t b = chunk_begin[c];
  const int32_t row_end = t_indptr[row + 1];
  const int32_t e = (b + T_CHUNK < row_end) ? b + T_CHUNK : row_end;
  double acc0 = 0., acc1 = 0.;
  int32_t k = b + lane;
  for (; k + WAVE < e; k += 2 * WAVE) {
    const double x0 = w[t_indices[k]];
    const double x1 = w[t_indices[k + WAVE]];
    acc0 += BIN ? x0 : t_data[k] * x0;
    acc1 += BIN ? x1 : t_data[k + WAVE] * x1;
  }
  if (k < e) {
    const double x0 = w[t_indices[k]];
    acc0 += BIN ? x0 : t_data[k] * x0;
  }
  const double tot = wave_sum(acc0 + acc1);
  if (lane == 0) partial[c] = tot;
}

__device__ inline double block_sum_256(double x) {
  __shared__ double s_w[256 / WAVE];
  x = wave_sum(x);
  if ((threadIdx.x & (WAVE - 1)) == 0) s_w[threadIdx.x / WAVE] = x;
  __syncthreads();
  double r = 0.;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 256 / WAVE; ++k) r += s_w[k];
  }
  __syncthreads();
  return r;  // valid in thread 0
}

// Adds the chunk partials of each row of X^T in chunk order, applies the
// intercept/centring correction and one of the CG epilogues.  Grid = NPART.
// `gfull` != nullptr means the main product comes as n_slab partial slabs of
// p entries each (the tiled and dense paths), added here in slab order; then
// row_chunk_ptr/partial are unused.
// (the epilogue mode is a template parameter: the kernel runs twice per CG
// iteration and is pure latency, so the branches and loads of the modes it is
// not in are worth compiling away)
// FOLD (TD_OPER_UPD / TD_RESID): the new residual also goes out scaled, s.*r,
// with the partials of <offset, (s.*r)[1:]> -- what the folded direction step
// of the next X~ v kernel starts from (common.hpp DotFold).
template <int mode, bool FOLD = false>
__global__ __launch_bounds__(VEC_BLOCK) void tdot_finalize_kernel(
    int64_t p, int intercept, const int32_t* __restrict__ row_chunk_ptr,
    const double* __restrict__ partial, const double* __restrict__ gfull,
    int n_slab, int64_t slab_stride, const double* __restrict__ offset,
    const double* __restrict__ sumw_part,
    const double* __restrict__ s, const double* __restrict__ d,
    const double* __restrict__ x, const double* __restrict__ z,
    const double* __restrict__ phi, const double* __restrict__ eta2,
    double* __restrict__ out, double* __restrict__ dot_part,
    // TD_OPER_UPD (x is the search direction p):
    double* __restrict__ cg_x, double* __restrict__ cg_r,
    CGState* __restrict__ cg_state, int cg_k,
    const double* __restrict__ pdp_part, const double* __restrict__ twt_part,
    double* __restrict__ fold_sr, double* __restrict__ fold_cr_part) {
  const int64_t P = p + intercept;
  const int64_t stride = (int64_t)gridDim.x * VEC_BLOCK;
  const int64_t jj0 = (int64_t)blockIdx.x * VEC_BLOCK + threadIdx.x;
  // This kernel is pure latency (400 KB of output): every load that depends
  // on nothing else -- the sum(w) partials, the first 16 slabs, the centring
  // offset and the epilogue vectors of the thread's first element -- is issued
  // before the first wait, one memory round trip instead of three.
  constexpr int PRE = 16;
  double pw[NPART / WAVE];
#pragma unroll
  for (int k = 0; k < NPART / WAVE; ++k)
    pw[k] = sumw_part[(threadIdx.x & (WAVE - 1)) + k * WAVE];  // branch-free
  const bool has0 = jj0 < P;
  const bool main0 = has0 && !(intercept && jj0 == 0);
  const int64_t jm0 = jj0 - intercept;
  double sv[PRE];
#pragma unroll
  for (int u = 0; u < PRE; ++u)
    sv[u] = (main0 && gfull && u < n_slab)
                ? gfull[(int64_t)u * slab_stride + jm0] : 0.;
  const double off0 = main0 ? offset[jm0] : 0.;
  double e0 = 0., e1 = 0., e2 = 0., e3 = 0.;
  // TD_OPER_UPD: the two halves of the curvature p.Ap and the CG scalars
  double pa[NPART / WAVE], pb[NPART / WAVE];
  double rho = 0.;
  if (mode == TD_OPER_UPD) {
#pragma unroll
    for (int k = 0; k < NPART / WAVE; ++k) {
      pa[k] = pdp_part[(threadIdx.x & (WAVE - 1)) + k * WAVE];
      pb[k] = twt_part[(threadIdx.x & (WAVE - 1)) + k * WAVE];
    }
    rho = cg_state->rho[cg_k & 1];
    if (cg_state->done) return;  // the stop rule has fired (uniform)
  }
  if (has0) {
    if (mode == TD_OPER || mode == TD_OPER_UPD) {
      e0 = x[jj0];
      e1 = d[jj0];
      e2 = s[jj0];
      if (mode == TD_OPER_UPD) e3 = cg_r[jj0];
    } else if (mode == TD_RESID) {
      e0 = z[jj0];
      e1 = phi[jj0];
      e2 = eta2[jj0];
      e3 = s[jj0];
    }
  }
  // sum(w): same adds in the same order as sum_partials()
  __shared__ double s_sumw, s_pap;
  if (threadIdx.x < WAVE) {
    double a = 0.;
#pragma unroll
    for (int k = 0; k < NPART / WAVE; ++k) a += pw[k];
    a = wave_sum(a);
    if (threadIdx.x == 0) s_sumw = a;
    if (mode == TD_OPER_UPD) {
      // <p, d p> and <t, Omega t>, each summed like sum_partials(), then added
      double b = 0., c = 0.;
#pragma unroll
      for (int k = 0; k < NPART / WAVE; ++k) {
        b += pa[k];
        c += pb[k];
      }
      b = wave_sum(b);
      c = wave_sum(c);
      if (threadIdx.x == 0) s_pap = b + c;
    }
  }
  __syncthreads();
  const double sumw = s_sumw;
  const double alpha = (mode == TD_OPER_UPD) ? rho / s_pap : 0.;
  double dacc = 0., cacc = 0.;
  for (int64_t jj = jj0; jj < P; jj += stride) {
    const bool first = jj == jj0;
    double g;
    double off_j = 0.;   // centring offset of this coordinate (0 for the intercept)
    if (intercept && jj == 0) {
      g = sumw;
    } else {
      const int64_t j = jj - intercept;
      if (gfull) {
        // partial slabs, added in slab order
        g = 0.;
        int k = 0;
        if (first) {
#pragma unroll
          for (int u = 0; u < PRE; ++u)
            if (u < n_slab) g += sv[u];
          k = n_slab < PRE ? n_slab : PRE;
        }
        for (; k + 8 <= n_slab; k += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            v[u] = gfull[(int64_t)(k + u) * slab_stride + j];
#pragma unroll
          for (int u = 0; u < 8; ++u) g += v[u];
        }
        for (; k < n_slab; ++k) g += gfull[(int64_t)k * slab_stride + j];
      } else {
        g = 0.;
        const int32_t cb = row_chunk_ptr[j], ce = row_chunk_ptr[j + 1];
        for (int32_t c = cb; c < ce; ++c) g += partial[c];
      }
      off_j = first ? off0 : offset[j];
      g -= sumw * off_j;
    }
    double r;
    if (mode == TD_OPER) {
      const double xj = first ? e0 : x[jj];
      r = (first ? e1 : d[jj]) * xj + (first ? e2 : s[jj]) * g;
      dacc += xj * r;
    } else if (mode == TD_OPER_UPD) {
      const double pj = first ? e0 : x[jj];
      const double q = (first ? e1 : d[jj]) * pj + (first ? e2 : s[jj]) * g;
      cg_x[jj] += alpha * pj;
      r = (first ? e3 : cg_r[jj]) - alpha * q;
      cg_r[jj] = r;
      dacc += r * r;
      if (FOLD) {
        const double srj = (first ? e2 : s[jj]) * r;
        fold_sr[jj] = srj;
        cacc += off_j * srj;
      }
      continue;
    } else if (mode == TD_RESID) {
      r = (first ? e3 : s[jj]) *
          ((first ? e0 : z[jj]) +
           ((first ? e1 : phi[jj]) * (first ? e2 : eta2[jj]) - g));
      if (x) r -= d[jj] * x[jj];
      dacc += r * r;
      if (FOLD) {
        const double srj = (first ? e3 : s[jj]) * r;
        fold_sr[jj] = srj;
        cacc += off_j * srj;
      }
    } else {
      r = g;
    }
    out[jj] = r;
  }
  if (dot_part) {
    const double tot = block_sum_256(dacc);
    if (threadIdx.x == 0) dot_part[blockIdx.x] = tot;
  }
  if (FOLD) {
    const double tot = block_sum_256(cacc);
    if (threadIdx.x == 0) fold_cr_part[blockIdx.x] = tot;
  }
  if (mode == TD_OPER_UPD && blockIdx.x == 0 && threadIdx.x == 0)
    cg_state->n_iter = cg_k + 1;
}

// ------------------------------------------------------------------ launches

static int pick_group(double mean_row_nnz) {
  if (mean_row_nnz <= 6.) return 4;
  if (mean_row_nnz <= 12.) return 8;
  if (mean_row_nnz <= 24.) return 16;
  if (mean_row_nnz <= 96.) return 32;
  return 64;
}

template <bool BIN>
static void launch_csr_dot_g(bbx_design* h, int G, int grid, const double* d_v,
                             const double* d_rowscale, double* d_t) {
  const int32_t* ip = h->indptr.as<int32_t>();
  const int32_t* ix = h->indices.as<int32_t>();
  const double* da = h->data.as<double>();
  const double* cp = part_slot(h, PS_C);
#define BBX_LAUNCH_G(GG)                                                       \
  BBX_LAUNCH((csr_dot_kernel<GG, BIN>), dim3(grid), dim3(256), 0,      \
                     h->stream, h->n, ip, ix, da, d_v, h->intercept, cp,       \
                     d_rowscale, d_t, h->skip_flag)
  switch (G) {
    case 4: BBX_LAUNCH_G(4); break;
    case 8: BBX_LAUNCH_G(8); break;
    case 16: BBX_LAUNCH_G(16); break;
    case 32: BBX_LAUNCH_G(32); break;
    default: BBX_LAUNCH_G(64); break;
  }
#undef BBX_LAUNCH_G
}

int launch_dot_csr(bbx_design* h, const double* d_v, const double* d_rowscale,
                   double* d_t) {
  const int G = pick_group(h->n > 0 ? (double)h->nnz / (double)h->n : 1.);
  const int64_t rows_per_block = 256 / G;
  int64_t nb = (h->n + rows_per_block - 1) / rows_per_block;
  if (nb > 256 * 64) nb = 256 * 64;
  if (nb < 1) nb = 1;
  BBX_TRY(timer_begin(h, 0));
  if (h->binary)
    launch_csr_dot_g<true>(h, G, (int)nb, d_v, d_rowscale, d_t);
  else
    launch_csr_dot_g<false>(h, G, (int)nb, d_v, d_rowscale, d_t);
  BBX_TRY(timer_end(h, 0));
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

#define BBX_FINALIZE_MODES(LAUNCH)                                             \
  if (ep.fold_sr && ep.mode != TD_OPER_UPD && ep.mode != TD_RESID)             \
    return fail(BBX_ERR_INVALID, "fold outputs need TD_OPER_UPD or TD_RESID"); \
  switch (ep.mode) {                                                           \
    case TD_PLAIN: LAUNCH(TD_PLAIN, false); break;                             \
    case TD_OPER: LAUNCH(TD_OPER, false); break;                               \
    case TD_OPER_UPD:                                                          \
      if (ep.fold_sr) LAUNCH(TD_OPER_UPD, true);                               \
      else LAUNCH(TD_OPER_UPD, false);                                         \
      break;                                                                   \
    case TD_RESID:                                                             \
      if (ep.fold_sr) LAUNCH(TD_RESID, true);                                  \
      else LAUNCH(TD_RESID, false);                                            \
      break;                                                                   \
    default: return fail(BBX_ERR_INVALID, "unknown Tdot epilogue mode");       \
  }

int launch_tdot_finalize(bbx_design* h, const double* d_gfull, int n_slab,
                         const double* d_sumw_part, const TdotEpilogue& ep,
                         double* d_out) {
#define BBX_FIN(MODE, FF)                                                      \
  BBX_LAUNCH((tdot_finalize_kernel<MODE, FF>), dim3(NPART),            \
                     dim3(VEC_BLOCK), 0, h->stream, h->p, h->intercept,        \
                     h->t_row_chunk_ptr.as<int32_t>(),                         \
                     h->t_partial.as<double>(), d_gfull, n_slab, h->p,         \
                     h->offset.as<double>(), d_sumw_part, ep.s, ep.d, ep.x,    \
                     ep.z, ep.phi, ep.eta2, d_out, ep.dot_part, ep.cg_x,       \
                     ep.cg_r, ep.cg_state, ep.cg_k, ep.pdp_part, ep.twt_part,  \
                     ep.fold_sr, ep.fold_cr_part)
  BBX_FINALIZE_MODES(BBX_FIN)
#undef BBX_FIN
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

// Dense operator: intercept column and centring are part of the matrix, so the
// epilogue sees a plain P-column product (intercept 0, offset 0, sum(w) 0).
int launch_tdot_finalize_dense(bbx_design* h, const TdotEpilogue& ep,
                               double* d_out, const double* d_slab,
                               int n_slab) {
#define BBX_FIN(MODE, FF)                                                      \
  BBX_LAUNCH((tdot_finalize_kernel<MODE, FF>), dim3(NPART),            \
                     dim3(VEC_BLOCK), 0, h->stream, h->P, 0, nullptr, nullptr, \
                     d_slab ? d_slab : h->dense_slab.as<double>(),             \
                     d_slab ? n_slab : h->dense_chunks, h->dense_ld,           \
                     h->offset.as<double>(), part_slot(h, PS_ZERO), ep.s,      \
                     ep.d, ep.x, ep.z, ep.phi, ep.eta2, d_out, ep.dot_part,    \
                     ep.cg_x, ep.cg_r, ep.cg_state, ep.cg_k, ep.pdp_part,      \
                     ep.twt_part, ep.fold_sr, ep.fold_cr_part)
  BBX_FINALIZE_MODES(BBX_FIN)
#undef BBX_FIN
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

static int launch_tdot_chunks_csr(bbx_design* h, const double* d_w) {
  BBX_TRY(timer_begin(h, 1));
  if (h->n_tchunk > 0) {
    const int64_t nb = (h->n_tchunk + 3) / 4;
    if (h->binary)
      BBX_LAUNCH(csr_tdot_chunk_kernel<true>, dim3((unsigned)nb),
                         dim3(256), 0, h->stream, h->n_tchunk,
                         h->t_chunk_row.as<int32_t>(),
                         h->t_chunk_begin.as<int32_t>(),
                         h->t_indptr.as<int32_t>(), h->t_indices.as<int32_t>(),
                         h->t_data.as<double>(), d_w,
                         h->t_partial.as<double>(), h->skip_flag);
    else
      BBX_LAUNCH(csr_tdot_chunk_kernel<false>, dim3((unsigned)nb),
                         dim3(256), 0, h->stream, h->n_tchunk,
                         h->t_chunk_row.as<int32_t>(),
                         h->t_chunk_begin.as<int32_t>(),
                         h->t_indptr.as<int32_t>(), h->t_indices.as<int32_t>(),
                         h->t_data.as<double>(), d_w,
                         h->t_partial.as<double>(), h->skip_flag);
  }
  BBX_TRY(timer_end(h, 1));
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_tdot_csr(bbx_design* h, const double* d_w,
                    const double* d_sumw_part, const TdotEpilogue& ep,
                    double* d_out) {
  BBX_TRY(launch_tdot_chunks_csr(h, d_w));
  return launch_tdot_finalize(h, nullptr, 0, d_sumw_part, ep, d_out);
}

// ------------------------------------------------------ transpose at set-up

__global__ void expand_rows_kernel(int64_t n, const int32_t* __restrict__ indptr,
                                   int32_t* __restrict__ rowid) {
  // one wave per row
  const int lane = threadIdx.x & (WAVE - 1);
  for (int64_t row = (int64_t)blockIdx.x * (blockDim.x / WAVE) +
                     threadIdx.x / WAVE;
       row < n; row += (int64_t)gridDim.x * (blockDim.x / WAVE)) {
    const int32_t b = indptr[row], e = indptr[row + 1];
    for (int32_t k = b + lane; k < e; k += WAVE) rowid[k] = (int32_t)row;
  }
}

__global__ void iota_kernel(int64_t n, int32_t* __restrict__ a) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    a[i] = (int32_t)i;
}

// t_indptr[j] = first position whose sorted column id is >= j.
__global__ void lower_bound_kernel(int64_t p, int64_t nnz,
                                   const int32_t* __restrict__ sorted_col,
                                   int32_t* __restrict__ t_indptr) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > p) return;
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_col[mid] < j) lo = mid + 1; else hi = mid;
  }
  t_indptr[j] = (int32_t)lo;
}

__global__ void gather_perm_kernel(int64_t nnz, const int32_t* __restrict__ perm,
                                   const int32_t* __restrict__ rowid,
                                   const double* __restrict__ data,
                                   int32_t* __restrict__ t_indices,
                                   double* __restrict__ t_data) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t k = perm[i];
    t_indices[i] = rowid[k];
    if (t_data) t_data[i] = data[k];
  }
}

// Builds CSR of X^T on the device with one stable radix sort by column id
// (rows stay ascending inside every column because CSR order is row-major),
// then the chunk table that load-balances the skewed columns.
int build_transpose_csr(bbx_design* h) {
  const int64_t nnz = h->nnz, p = h->p, n = h->n;
  BBX_TRY(h->t_indptr.alloc(sizeof(int32_t) * (size_t)(p + 1)));
  BBX_TRY(h->t_indices.alloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
  if (!h->binary)
    BBX_TRY(h->t_data.alloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1)));
  std::vector<int32_t> t_indptr((size_t)p + 1, 0);
  if (nnz > 0) {
    DevMem rowid, perm_in, perm_out, col_sorted, tmp;
    BBX_TRY(rowid.alloc(sizeof(int32_t) * (size_t)nnz));
    BBX_TRY(perm_in.alloc(sizeof(int32_t) * (size_t)nnz));
    BBX_TRY(perm_out.alloc(sizeof(int32_t) * (size_t)nnz));
    BBX_TRY(col_sorted.alloc(sizeof(int32_t) * (size_t)nnz));
    BBX_LAUNCH(expand_rows_kernel, dim3(2048), dim3(256), 0, h->stream,
                       n, h->indptr.as<int32_t>(), rowid.as<int32_t>());
    BBX_LAUNCH(iota_kernel, dim3(2048), dim3(256), 0, h->stream, nnz,
                       perm_in.as<int32_t>());
    int end_bit = 1;
    while (end_bit < 31 && ((int64_t)1 << end_bit) < p) ++end_bit;
    size_t tmp_bytes = 0;
    BBX_HIP(hipcub::DeviceRadixSort::SortPairs(
        nullptr, tmp_bytes, h->indices.as<int32_t>(), col_sorted.as<int32_t>(),
        perm_in.as<int32_t>(), perm_out.as<int32_t>(), (int)nnz, 0, end_bit,
        h->stream));
    BBX_TRY(tmp.alloc(tmp_bytes > 0 ? tmp_bytes : 1));
    BBX_HIP(hipcub::DeviceRadixSort::SortPairs(
        tmp.ptr, tmp_bytes, h->indices.as<int32_t>(), col_sorted.as<int32_t>(),
        perm_in.as<int32_t>(), perm_out.as<int32_t>(), (int)nnz, 0, end_bit,
        h->stream));
    BBX_LAUNCH(gather_perm_kernel, dim3(2048), dim3(256), 0, h->stream,
                       nnz, perm_out.as<int32_t>(), rowid.as<int32_t>(),
                       h->data.as<double>(), h->t_indices.as<int32_t>(),
                       h->binary ? nullptr : h->t_data.as<double>());
    BBX_LAUNCH(lower_bound_kernel, dim3((unsigned)((p + 256) / 256)),
                       dim3(256), 0, h->stream, p, nnz,
                       col_sorted.as<int32_t>(), h->t_indptr.as<int32_t>());
    BBX_HIP(hipGetLastError());
    BBX_HIP(hipMemcpyAsync(t_indptr.data(), h->t_indptr.ptr,
                           sizeof(int32_t) * (size_t)(p + 1),
                           hipMemcpyDeviceToHost, h->stream));
    BBX_HIP(hipStreamSynchronize(h->stream));
  } else {
    BBX_HIP(hipMemsetAsync(h->t_indptr.ptr, 0,
                           sizeof(int32_t) * (size_t)(p + 1), h->stream));
  }
  // chunk table (host; p + nnz/T_CHUNK entries)
  std::vector<int32_t> chunk_row, chunk_begin, row_chunk_ptr((size_t)p + 1);
  chunk_row.reserve((size_t)(p + nnz / T_CHUNK + 1));
  chunk_begin.reserve((size_t)(p + nnz / T_CHUNK + 1));
  for (int64_t j = 0; j < p; ++j) {
    row_chunk_ptr[(size_t)j] = (int32_t)chunk_row.size();
    for (int32_t b = t_indptr[(size_t)j]; b < t_indptr[(size_t)j + 1];
         b += T_CHUNK) {
      chunk_row.push_back((int32_t)j);
      chunk_begin.push_back(b);
    }
  }
  row_chunk_ptr[(size_t)p] = (int32_t)chunk_row.size();
  h->n_tchunk = (int64_t)chunk_row.size();
  const size_t nc = chunk_row.size() > 0 ? chunk_row.size() : 1;
  BBX_TRY(h->t_chunk_row.alloc(sizeof(int32_t) * nc));
  BBX_TRY(h->t_chunk_begin.alloc(sizeof(int32_t) * nc));
  BBX_TRY(h->t_row_chunk_ptr.alloc(sizeof(int32_t) * (size_t)(p + 1)));
  BBX_TRY(h->t_partial.alloc(sizeof(double) * nc));
  if (!chunk_row.empty()) {
    BBX_HIP(hipMemcpy(h->t_chunk_row.ptr, chunk_row.data(),
                      sizeof(int32_t) * chunk_row.size(),
                      hipMemcpyHostToDevice));
    BBX_HIP(hipMemcpy(h->t_chunk_begin.ptr, chunk_begin.data(),
                      sizeof(int32_t) * chunk_begin.size(),
                      hipMemcpyHostToDevice));
  }
  BBX_HIP(hipMemcpy(h->t_row_chunk_ptr.ptr, row_chunk_ptr.data(),
                    sizeof(int32_t) * row_chunk_ptr.size(),
                    hipMemcpyHostToDevice));
  return BBX_OK;
}

}  // namespace bbx
