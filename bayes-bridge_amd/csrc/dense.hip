// Dense design operator: row-major n x ld matrix in HBM (f32 or f64 storage,
// f64 arithmetic), with the centring and the intercept column applied once
// while the device copy is made.
//
// Replaces DenseDesignMatrix.__init__/dot/Tdot (design_matrix/dense_matrix.py:
// 9-27,37-52), whose arithmetic is NumPy/OpenBLAS dgemv.  Both products are
// HBM-bound streams of the matrix (0.25 flop/byte at f64, 0.5 at f32 storage):
//   dot : one wavefront per row, 16-byte lane loads, shuffle reduction;
//   Tdot: a thread owns 4 adjacent columns and walks a chunk of rows; the
//         per-chunk partial sums form slabs that the common Tdot epilogue adds
//         in chunk order (no atomics, bitwise reproducible).
#include <new>
#include <vector>

#include "common.hpp"

namespace bbx {

// The matrix is read once per product: non-temporal loads keep it from evicting
// the vectors and slabs the kernels re-read (same idea as the id stream of the
// tiled layout).
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef double nt_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 stream_load(const float4* p) {
  const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double4 stream_load(const double4* p) {
  const nt_d2* q = reinterpret_cast<const nt_d2*>(p);
  const nt_d2 a = __builtin_nontemporal_load(q);
  const nt_d2 b = __builtin_nontemporal_load(q + 1);
  return make_double4(a.x, a.y, b.x, b.y);
}

constexpr int DENSE_ROW_CHUNKS = 64;

template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
  using type = float4;
};
template <>
struct Vec4<double> {
  using type = double4;
};

// out[i] = rowscale[i] * sum_j X[i, j] v[j]
// The vector (ld doubles, <= 64 KB at P = 8001) is staged in LDS once per
// workgroup and reused for every row it handles; one wavefront per row streams
// the row with 16-byte lane loads, four in flight.
template <typename T>
__global__ __launch_bounds__(256) void dense_dot_kernel(
    int64_t n, int64_t P, int64_t ld, const T* __restrict__ X,
    const double* __restrict__ v, const double* __restrict__ rowscale,
    double* __restrict__ out, const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;  // the CG solve has already stopped
  using V4 = typename Vec4<T>::type;
  extern __shared__ __attribute__((aligned(16))) double vs[];
  for (int64_t j = threadIdx.x; j < ld; j += 256) vs[j] = (j < P) ? v[j] : 0.;
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave0 = (int64_t)blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  const int64_t n_wave = (int64_t)gridDim.x * (256 / WAVE);
  const int64_t nq = ld / 4;
  const double4* __restrict__ v4 = reinterpret_cast<const double4*>(vs);
  for (int64_t row = wave0; row < n; row += n_wave) {
    const V4* __restrict__ xr = reinterpret_cast<const V4*>(X + row * ld);
    double a0 = 0., a1 = 0.;
    int64_t q = lane;
    for (; q + 3 * WAVE < nq; q += 4 * WAVE) {
      const V4 x0 = stream_load(xr + q), x1 = stream_load(xr + q + WAVE),
               x2 = stream_load(xr + q + 2 * WAVE),
               x3 = stream_load(xr + q + 3 * WAVE);
      const double4 w0 = v4[q], w1 = v4[q + WAVE], w2 = v4[q + 2 * WAVE],
                    w3 = v4[q + 3 * WAVE];
      a0 += (double)x0.x * w0.x + (double)x0.z * w0.z;
      a1 += (double)x0.y * w0.y + (double)x0.w * w0.w;
      a0 += (double)x1.x * w1.x + (double)x1.z * w1.z;
      a1 += (double)x1.y * w1.y + (double)x1.w * w1.w;
      a0 += (double)x2.x * w2.x + (double)x2.z * w2.z;
      a1 += (double)x2.y * w2.y + (double)x2.w * w2.w;
      a0 += (double)x3.x * w3.x + (double)x3.z * w3.z;
      a1 += (double)x3.y * w3.y + (double)x3.w * w3.w;
    }
    for (; q < nq; q += WAVE) {
      const V4 x0 = stream_load(xr + q);
      const double4 w0 = v4[q];
      a0 += (double)x0.x * w0.x + (double)x0.z * w0.z;
      a1 += (double)x0.y * w0.y + (double)x0.w * w0.w;
    }
    double a = wave_allsum(a0 + a1);
    if (lane == 0) {
      if (rowscale) a *= rowscale[row];
      out[row] = a;
    }
  }
}

// The same GEMV on the matrix cores (opt-in, BBX_DENSE_MFMA=1): the A/B
// measurement BASELINE.json's "MFMA-tiled GEMV for dense X" asks for.
// v_mfma_f64_16x16x4_f64 computes D[16x16] += A[16x4] B[4x16]; a GEMV has ONE
// right-hand side, so the 16 columns of B all carry the same slice of v and 15
// of the 16 result columns are redundant (there is no way to give the columns
// different k-slices: A is shared by all of them).  A wave owns 16 rows; lane
// (i = l & 15, k = l >> 4) loads 16 bytes X[i][16c + 4k .. 4k+3] (the four
// lanes of a row cover one 64-byte segment), converts to f64 and feeds four
// MFMAs per load.  Operand maps (cdna_hip_programming.md, "f64 MFMA"):
// A[l&15][l>>4], B[l>>4][l&15], D col = l & 15, row = (l >> 4) + 4 reg.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void dense_dot_mfma_kernel(
    int64_t n, int64_t P, int64_t ld, const float* __restrict__ X,
    const double* __restrict__ v, const double* __restrict__ rowscale,
    double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double vs[];
  for (int64_t j = threadIdx.x; j < ld + 16; j += 256) vs[j] = (j < P) ? v[j] : 0.;
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1);
  const int i = lane & 15, k = lane >> 4;
  const int64_t blk0 = (int64_t)blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  const int64_t n_wave = (int64_t)gridDim.x * (256 / WAVE);
  const int64_t n_blk = (n + 15) / 16;
  const int64_t n_chunk = (ld + 15) / 16;
  for (int64_t blk = blk0; blk < n_blk; blk += n_wave) {
    int64_t row = blk * 16 + i;
    const bool live = row < n;
    if (!live) row = n - 1;
    const float* __restrict__ xr = X + row * ld + 4 * k;
    // four independent accumulators: back-to-back MFMAs on ONE accumulator
    // serialise on its 8-pass latency (measured: 1.67 ms instead of the rate
    // below); they are added once at the end
    mfma_d4 acc = {0., 0., 0., 0.}, acc1 = acc, acc2 = acc, acc3 = acc;
    int64_t c = 0;
    for (; c + 3 < n_chunk; c += 4) {
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t col = 16 * (c + u) + 4 * k;
        x[u] = (col < ld) ? stream_load(reinterpret_cast<const float4*>(
                                xr + 16 * (c + u)))
                          : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double* b = vs + 16 * (c + u) + 4 * k;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x[u].x, b[0], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x[u].y, b[1], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x[u].z, b[2], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x[u].w, b[3], acc3, 0, 0, 0);
      }
    }
    for (; c < n_chunk; ++c) {
      const int64_t col = 16 * c + 4 * k;
      const float4 x0 = (col < ld) ? stream_load(reinterpret_cast<const float4*>(
                                         xr + 16 * c))
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
      const double* b = vs + col;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x0.x, b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x0.y, b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x0.z, b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)x0.w, b[3], acc, 0, 0, 0);
    }
    acc = (acc + acc1) + (acc2 + acc3);
    // every column of D holds the result; column 0 lives in lanes 0, 16, 32, 48
    if (i == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t orow = blk * 16 + k + 4 * r;
        if (orow < n) {
          double a = acc[r];
          if (rowscale) a *= rowscale[orow];
          out[orow] = a;
        }
      }
    }
  }
}

// slab[chunk][j] = sum over the chunk's rows of X[i, j] w[i]
template <typename T>
__global__ __launch_bounds__(256) void dense_tdot_kernel(
    int64_t n, int64_t ld, int64_t rows_per_chunk, const T* __restrict__ X,
    const double* __restrict__ w, double* __restrict__ slab,
    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;  // the CG solve has already stopped
  using V4 = typename Vec4<T>::type;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // column quad
  if (q * 4 >= ld) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > n) r1 = n;
  double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
  const V4* __restrict__ xq = reinterpret_cast<const V4*>(X) + q;
  const int64_t ldq = ld / 4;
  int64_t i = r0;
  for (; i + 1 < r1; i += 2) {
    const V4 xa = stream_load(xq + i * ldq);
    const V4 xb = stream_load(xq + (i + 1) * ldq);
    const double wa = w[i], wb = w[i + 1];
    a0 += (double)xa.x * wa;
    a1 += (double)xa.y * wa;
    a2 += (double)xa.z * wa;
    a3 += (double)xa.w * wa;
    a0 += (double)xb.x * wb;
    a1 += (double)xb.y * wb;
    a2 += (double)xb.z * wb;
    a3 += (double)xb.w * wb;
  }
  if (i < r1) {
    const V4 xa = stream_load(xq + i * ldq);
    const double wa = w[i];
    a0 += (double)xa.x * wa;
    a1 += (double)xa.y * wa;
    a2 += (double)xa.z * wa;
    a3 += (double)xa.w * wa;
  }
  double* dst = slab + (int64_t)blockIdx.y * ld + q * 4;
  dst[0] = a0;
  dst[1] = a1;
  dst[2] = a2;
  dst[3] = a3;
}

// ---- one pass over X for  g = X^T (rowscale .* (X v)) ----------------------
//
// A CG iteration on a dense design reads the matrix twice (dot, then Tdot):
// 2 x 6.4 GB at 200k x 8k.  Here a 1024-thread workgroup owns a contiguous
// range of rows and walks it RB rows at a time.  A thread owns KQ groups of 4
// adjacent columns (its slice of v and of the result stays in registers); the
// RB x 4 KQ matrix entries it loaded stay in registers between the two
// products:
//   p_i = <X[i, own cols], v[own cols]>       lane-private
//   t_i = sum over the 1024 threads of p_i     shuffles, LDS, ONE barrier
//   g[own cols] += X[i, own cols] * (rowscale_i * t_i)
// Per-workgroup results go to slabs that the common Tdot epilogue adds in
// workgroup order (fixed order, no atomics), and sum_i rowscale_i t_i^2 -- the
// data half of the CG curvature p.Ap, see cg_sampler.hip -- to one partial
// per workgroup.  Two kernels share the arithmetic below (explicit fma, so that
// they agree bit for bit whatever the compiler would contract):
//   dense_fused_kernel       the next RB rows are prefetched into REGISTERS
//                            before the reduction starts (f32 and f64 storage);
//   dense_fused_ring_kernel  the stream goes through an LDS ring filled by
//                            LDS-DMA, D blocks ahead (f32 storage).
template <typename V4, int KQ>
__device__ __forceinline__ double fused_row_dot(const V4 (&x)[KQ],
                                                const double4 (&vo)[KQ]) {
  double a0 = 0., a1 = 0.;
#pragma unroll
  for (int k = 0; k < KQ; ++k) {
    a0 = fma((double)x[k].x, vo[k].x, a0);
    a1 = fma((double)x[k].y, vo[k].y, a1);
    a0 = fma((double)x[k].z, vo[k].z, a0);
    a1 = fma((double)x[k].w, vo[k].w, a1);
  }
  return a0 + a1;  // this lane's part of the row's inner product
}

// The row sums cross lanes without LDS traffic (common.hpp: row16_allsum,
// lane_value), plus gfx950's v_permlane32_swap between the halves of the
// wavefront.  With __shfl_down (two ds_bpermute_b32 per step and double: 24
// LDS round trips per block of two rows, on the critical path in front of the
// workgroup barrier) the ring kernel ran 1.17 ms at 200k x 8k; with these 0.92.
// Wave totals of two lane-private values at once (uniform results): after the
// swap lanes 0-31 hold p0[l] + p0[l+32] and lanes 32-63 p1[l-32] + p1[l]; one
// 16-lane all-sum then serves both rows.  Fixed order.
typedef unsigned fused_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wave_sum_pair(double p0, double p1, double& t0,
                                              double& t1) {
  const fused_v2u lo = __builtin_amdgcn_permlane32_swap(
      (unsigned)__double2loint(p0), (unsigned)__double2loint(p1), false, false);
  const fused_v2u hi = __builtin_amdgcn_permlane32_swap(
      (unsigned)__double2hiint(p0), (unsigned)__double2hiint(p1), false, false);
  const double m = row16_allsum(__hiloint2double((int)hi.x, (int)lo.x) +
                                __hiloint2double((int)hi.y, (int)lo.y));
  t0 = lane_value(m, 0) + lane_value(m, 16);
  t1 = lane_value(m, 32) + lane_value(m, 48);
}
// Totals of two rows over the workgroup's 16 waves: red2 = [row 0: 16 wave
// partials | row 1: 16 wave partials]; lanes 0-15 / 16-31 take one partial each
// (one LDS read per lane instead of 32), the 16-lane all-sum does the rest.
__device__ __forceinline__ void block_sum_pair(const double* red2, int lane,
                                               double& t0, double& t1) {
  const double m = row16_allsum(red2[lane & 31]);
  t0 = lane_value(m, 0);
  t1 = lane_value(m, 16);
}

template <typename V4, int KQ>
__device__ __forceinline__ void fused_row_axpy(const V4 (&x)[KQ], double wi,
                                               double4 (&g)[KQ]) {
#pragma unroll
  for (int k = 0; k < KQ; ++k) {
    g[k].x = fma((double)x[k].x, wi, g[k].x);
    g[k].y = fma((double)x[k].y, wi, g[k].y);
    g[k].z = fma((double)x[k].z, wi, g[k].z);
    g[k].w = fma((double)x[k].w, wi, g[k].w);
  }
}

template <typename T, int KQ, int RB>
__global__ __launch_bounds__(1024) void dense_fused_kernel(
    int64_t n, int64_t P, int64_t ld, int64_t rows_per_wg,
    const T* __restrict__ X, const double* __restrict__ v,
    const double* __restrict__ rowscale, double* __restrict__ slab,
    const int* __restrict__ skip_flag, double* __restrict__ twt_part,
    const double* __restrict__ addend) {
  if (skip_flag && *skip_flag) return;  // the CG solve has already stopped
  __shared__ double red[2][RB][1024 / WAVE];
  double twt = 0.;  // sum_i rowscale_i t_i^2 over this workgroup's rows
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  const int64_t ldq = ld / 4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = (r0 + rows_per_wg < n) ? r0 + rows_per_wg : n;
  double4 vo[KQ], g[KQ];
  bool has[KQ];
#pragma unroll
  for (int k = 0; k < KQ; ++k) {
    const int64_t q = tid + 1024 * k;
    has[k] = q < ldq;
    vo[k] = make_double4(0., 0., 0., 0.);
    g[k] = make_double4(0., 0., 0., 0.);
    if (has[k]) {
      const int64_t c = q * 4;
      vo[k].x = (c + 0 < P) ? v[c + 0] : 0.;
      vo[k].y = (c + 1 < P) ? v[c + 1] : 0.;
      vo[k].z = (c + 2 < P) ? v[c + 2] : 0.;
      vo[k].w = (c + 3 < P) ? v[c + 3] : 0.;
    }
  }
  using V4 = typename Vec4<T>::type;
  const V4* __restrict__ X4 = reinterpret_cast<const V4*>(X);
  V4 xc[RB][KQ], xn[RB][KQ];
  double sc[RB], sn[RB], ac[RB], an[RB];
  auto load_block = [&](int64_t r, V4 (&x)[RB][KQ], double (&s)[RB],
                        double (&a)[RB]) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const bool ok = r + i < r1;
      s[i] = ok ? (rowscale ? rowscale[r + i] : 1.) : 0.;
      a[i] = (ok && addend) ? addend[r + i] : 0.;
#pragma unroll
      for (int k = 0; k < KQ; ++k)
        if (ok && has[k]) {
          x[i][k] = stream_load(X4 + (r + i) * ldq + tid + 1024 * k);
        } else {
          x[i][k].x = x[i][k].y = x[i][k].z = x[i][k].w = (T)0;
        }
    }
  };
  load_block(r0, xc, sc, ac);
  int buf = 0;
  for (int64_t r = r0; r < r1; r += RB) {
    load_block(r + RB, xn, sn, an);  // in flight across the reduction below
    static_assert(RB == 2, "the row sums are exchanged two rows at a time");
    double t[RB];
    wave_sum_pair(fused_row_dot<V4, KQ>(xc[0], vo),
                  fused_row_dot<V4, KQ>(xc[1], vo), t[0], t[1]);
    if (lane == 0) {
      red[buf][0][wave] = t[0];
      red[buf][1][wave] = t[1];
    }
    __syncthreads();
    block_sum_pair(&red[buf][0][0], lane, t[0], t[1]);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const double wi = sc[i] * t[i];
      twt = fma(wi, t[i], twt);
      fused_row_axpy<V4, KQ>(xc[i], wi + ac[i], g);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      sc[i] = sn[i];
      ac[i] = an[i];
#pragma unroll
      for (int k = 0; k < KQ; ++k) xc[i][k] = xn[i][k];
    }
    buf ^= 1;  // the next round's partials go to the other buffer: one barrier
  }
  double4* __restrict__ dst =
      reinterpret_cast<double4*>(slab + (int64_t)blockIdx.x * ld);
#pragma unroll
  for (int k = 0; k < KQ; ++k)
    if (has[k]) dst[tid + 1024 * k] = g[k];
  if (twt_part && tid == 0) twt_part[blockIdx.x] = twt;
}

// ---- the same pass with the prefetch in LDS instead of registers -----------
//
// dense_fused_kernel keeps ONE block of RB rows in flight per thread (its
// registers hold the current block, the next one and the thread's slices of v
// and g: 114 VGPRs at KQ = 2, a deeper register prefetch spills), and every
// block ends at a workgroup barrier, i.e. waits for the SLOWEST of the
// workgroup's outstanding loads.  Here the stream goes through an LDS ring
// filled by LDS-DMA (global_load_lds_dwordx4, no register destination): D
// blocks of RB rows are in flight ahead of the one being computed, the
// registers hold only the current block (93 VGPRs).  A thread reads back
// exactly the bytes its own DMA instructions wrote (lane-linear image), so the
// only ordering the data needs is the issuing wave's counted vmcnt -- the one
// barrier per block is still the row-sum exchange, and it is a raw s_barrier:
// a __syncthreads() would be free to drain the DMAs.
// Measured at 200k x 8k (profiles/r02_ab_dense_fused.txt): 1.165 vs 1.195 ms
// on one box; what is left above the 1.07 ms floor of this launch shape (the
// same kernel with no arithmetic at all) is the reduction chain per block, not
// the bytes in flight.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

template <int KQ, int RB, int D>
__global__ __launch_bounds__(1024) void dense_fused_ring_kernel(
    int64_t n, int64_t P, int64_t ld, int64_t rows_per_wg,
    const float* __restrict__ X, const double* __restrict__ v,
    const double* __restrict__ rowscale, double* __restrict__ slab,
    const int* __restrict__ skip_flag, double* __restrict__ twt_part,
    const double* __restrict__ addend) {
  if (skip_flag && *skip_flag) return;  // the CG solve has already stopped
  double twt = 0.;  // sum_i rowscale_i t_i^2 over this workgroup's rows
  constexpr int SLOT_Q = KQ * 1024;     // 16-byte units per row slot
  constexpr int NWAVE = 1024 / WAVE;
  extern __shared__ __attribute__((aligned(16))) unsigned char fused_smem[];
  float4* ring = reinterpret_cast<float4*>(fused_smem);  // [D * RB][SLOT_Q]
  double* red = reinterpret_cast<double*>(fused_smem + (size_t)D * RB * SLOT_Q * 16);
  double* rs = red + 2 * RB * NWAVE;                     // [rows_per_wg + RB]
  double* ad = rs + rows_per_wg + 8;                     // [rows_per_wg + RB]
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int64_t ldq = ld / 4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = (r0 + rows_per_wg < n) ? r0 + rows_per_wg : n;
  double4 vo[KQ], g[KQ];
  bool has[KQ];
  int64_t qs[KQ];  // source quad of this thread (clamped: every lane issues)
#pragma unroll
  for (int k = 0; k < KQ; ++k) {
    const int64_t q = tid + 1024 * k;
    has[k] = q < ldq;
    qs[k] = has[k] ? q : ldq - 1;
    vo[k] = make_double4(0., 0., 0., 0.);
    g[k] = make_double4(0., 0., 0., 0.);
    if (has[k]) {
      const int64_t c = q * 4;
      vo[k].x = (c + 0 < P) ? v[c + 0] : 0.;
      vo[k].y = (c + 1 < P) ? v[c + 1] : 0.;
      vo[k].z = (c + 2 < P) ? v[c + 2] : 0.;
      vo[k].w = (c + 3 < P) ? v[c + 3] : 0.;
    }
  }
  double4* __restrict__ dst =
      reinterpret_cast<double4*>(slab + (int64_t)blockIdx.x * ld);
  if (r0 >= r1) {  // no rows: a zero slab
#pragma unroll
    for (int k = 0; k < KQ; ++k)
      if (has[k]) dst[tid + 1024 * k] = g[k];
    if (twt_part && tid == 0) twt_part[blockIdx.x] = 0.;
    return;
  }
  const int n_rows = (int)(r1 - r0);
  const int n_blk = (n_rows + RB - 1) / RB;
  for (int j = tid; j < n_blk * RB; j += 1024) {
    rs[j] = j < n_rows ? (rowscale ? rowscale[r0 + j] : 1.) : 0.;
    ad[j] = (j < n_rows && addend) ? addend[r0 + j] : 0.;
  }
  // every compiler-visible load is retired before the counted ring starts
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
  __syncthreads();
  const float4* __restrict__ X4 = reinterpret_cast<const float4*>(X);
  const unsigned ring_lds = (unsigned)(uintptr_t)ring;
  // Rows past the end are clamped to the last row (their scale is 0): every
  // wave issues exactly RB * KQ DMA instructions per block, which is what the
  // counted waits below assume.
  auto issue = [&](int b, int slot) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      int64_t row = r0 + (int64_t)b * RB + i;
      if (row >= r1) row = r1 - 1;
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        const unsigned dst_lds = __builtin_amdgcn_readfirstlane(
            ring_lds + (unsigned)(((slot * RB + i) * SLOT_Q + k * 1024 + wave * WAVE) * 16));
        glds16(X4 + row * ldq + qs[k], dst_lds);
      }
    }
  };
  constexpr int PER_BLOCK = RB * KQ;
#pragma unroll
  for (int b = 0; b < D; ++b)
    if (b < n_blk) issue(b, b);
  int slot = 0, buf = 0;
  for (int b = 0; b < n_blk; ++b) {
    // block b has landed once at most the younger blocks' DMAs are pending
    if (b + D <= n_blk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PER_BLOCK) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float4 xc[RB][KQ];
    double sc[RB], ac[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      sc[i] = rs[b * RB + i];
      ac[i] = ad[b * RB + i];
#pragma unroll
      for (int k = 0; k < KQ; ++k) {
        xc[i][k] = ring[(size_t)(slot * RB + i) * SLOT_Q + k * 1024 + tid];
        if (!has[k]) xc[i][k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // the slot's bytes are in registers: refill it D blocks ahead
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (b + D < n_blk) issue(b + D, slot);
    static_assert(RB == 2, "the row sums are exchanged two rows at a time");
    double t[RB];
    wave_sum_pair(fused_row_dot<float4, KQ>(xc[0], vo),
                  fused_row_dot<float4, KQ>(xc[1], vo), t[0], t[1]);
    if (lane == 0) {
      red[(buf * RB + 0) * NWAVE + wave] = t[0];
      red[(buf * RB + 1) * NWAVE + wave] = t[1];
    }
    // LDS writes visible, then the barrier; the DMAs in flight are NOT waited for
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    block_sum_pair(red + buf * RB * NWAVE, lane, t[0], t[1]);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const double wi = sc[i] * t[i];
      twt = fma(wi, t[i], twt);
      fused_row_axpy<float4, KQ>(xc[i], wi + ac[i], g);
    }
    buf ^= 1;
    slot = (slot + 1 == D) ? 0 : slot + 1;
  }
#pragma unroll
  for (int k = 0; k < KQ; ++k)
    if (has[k]) dst[tid + 1024 * k] = g[k];
  if (twt_part && tid == 0) twt_part[blockIdx.x] = twt;
}


// ---- the single pass for f64 STORAGE (the reference's own dtype) -------------
//
// Until round 3 f64 storage went through dense_fused_kernel<double, KQ, 2>: a
// thread owned 4 adjacent doubles per group, i.e. 32 bytes per lane and load --
// two 16-byte instructions whose lanes are 32 bytes apart (half-used lines per
// instruction) -- and the LDS-DMA ring had no room for two rows of 8 008
// doubles: 2.31 ms for 12.8 GB = 0.69 of peak, against 0.87 for f32 storage.
// Here a thread owns G groups of TWO adjacent doubles (one 16-byte unit per
// group, units 1024 apart): every load instruction of a wave reads 1 KiB of
// contiguous bytes, exactly the f32 kernels' access shape, and a ROW of 8 008
// doubles is what a two-row f32 block is -- 64 KB -- so the ring kernel runs
// with one row per stage (RB = 1, D = 2: 128 KB of the 160 KB LDS).
//   dense_fused_f64_kernel       next RB = 2 rows prefetched into registers
//                                (default: 1.85 ms = 0.87 of peak at 200k x 8k)
//   dense_fused_f64_ring_kernel  LDS-DMA ring, RB rows per stage, D stages
//                                (opt-in: 1.91 ms = 0.84; a barrier per row)
// Same arithmetic in both (explicit fma, the same lane-private sums, the same
// wave / workgroup reduction order): bit-identical results.
typedef double fused_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fused_d2 stream_load2(const fused_d2* p) {
  return __builtin_nontemporal_load(p);
}
template <int G>
__device__ __forceinline__ double fused_row_dot2(const fused_d2 (&x)[G],
                                                 const fused_d2 (&vo)[G]) {
  double a0 = 0., a1 = 0.;
#pragma unroll
  for (int k = 0; k < G; ++k) {
    a0 = fma(x[k].x, vo[k].x, a0);
    a1 = fma(x[k].y, vo[k].y, a1);
  }
  return a0 + a1;
}
template <int G>
__device__ __forceinline__ void fused_row_axpy2(const fused_d2 (&x)[G],
                                                double wi, fused_d2 (&g)[G]) {
#pragma unroll
  for (int k = 0; k < G; ++k) {
    g[k].x = fma(x[k].x, wi, g[k].x);
    g[k].y = fma(x[k].y, wi, g[k].y);
  }
}
// this thread's slices of v and g: pairs q = tid + 1024 k of the ld / 2 pairs
template <int G>
__device__ __forceinline__ void fused_f64_setup(int tid, int64_t ldp, int64_t P,
                                                const double* __restrict__ v,
                                                fused_d2 (&vo)[G],
                                                fused_d2 (&g)[G], bool (&has)[G]) {
#pragma unroll
  for (int k = 0; k < G; ++k) {
    const int64_t q = tid + 1024 * k;
    has[k] = q < ldp;
    vo[k] = fused_d2{0., 0.};
    g[k] = fused_d2{0., 0.};
    if (has[k]) {
      vo[k].x = (2 * q < P) ? v[2 * q] : 0.;
      vo[k].y = (2 * q + 1 < P) ? v[2 * q + 1] : 0.;
    }
  }
}

template <int G>
__global__ __launch_bounds__(1024) void dense_fused_f64_kernel(
    int64_t n, int64_t P, int64_t ld, int64_t rows_per_wg,
    const double* __restrict__ X, const double* __restrict__ v,
    const double* __restrict__ rowscale, double* __restrict__ slab,
    const int* __restrict__ skip_flag, double* __restrict__ twt_part,
    const double* __restrict__ addend) {
  if (skip_flag && *skip_flag) return;  // the CG solve has already stopped
  constexpr int RB = 2;
  __shared__ double red[2][RB][1024 / WAVE];
  double twt = 0.;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  const int64_t ldp = ld / 2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = (r0 + rows_per_wg < n) ? r0 + rows_per_wg : n;
  fused_d2 vo[G], g[G];
  bool has[G];
  fused_f64_setup<G>(tid, ldp, P, v, vo, g, has);
  const fused_d2* __restrict__ X2 = reinterpret_cast<const fused_d2*>(X);
  fused_d2 xc[RB][G], xn[RB][G];
  double sc[RB], sn[RB], ac[RB], an[RB];
  auto load_block = [&](int64_t r, fused_d2 (&x)[RB][G], double (&s)[RB],
                        double (&a)[RB]) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const bool ok = r + i < r1;
      s[i] = ok ? (rowscale ? rowscale[r + i] : 1.) : 0.;
      a[i] = (ok && addend) ? addend[r + i] : 0.;
#pragma unroll
      for (int k = 0; k < G; ++k)
        x[i][k] = (ok && has[k]) ? stream_load2(X2 + (r + i) * ldp + tid + 1024 * k)
                                 : fused_d2{0., 0.};
    }
  };
  load_block(r0, xc, sc, ac);
  int buf = 0;
  for (int64_t r = r0; r < r1; r += RB) {
    load_block(r + RB, xn, sn, an);  // in flight across the reduction below
    double t[RB];
    wave_sum_pair(fused_row_dot2<G>(xc[0], vo), fused_row_dot2<G>(xc[1], vo),
                  t[0], t[1]);
    if (lane == 0) {
      red[buf][0][wave] = t[0];
      red[buf][1][wave] = t[1];
    }
    __syncthreads();
    block_sum_pair(&red[buf][0][0], lane, t[0], t[1]);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const double wi = sc[i] * t[i];
      twt = fma(wi, t[i], twt);
      fused_row_axpy2<G>(xc[i], wi + ac[i], g);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      sc[i] = sn[i];
      ac[i] = an[i];
#pragma unroll
      for (int k = 0; k < G; ++k) xc[i][k] = xn[i][k];
    }
    buf ^= 1;
  }
  fused_d2* __restrict__ dst =
      reinterpret_cast<fused_d2*>(slab + (int64_t)blockIdx.x * ld);
#pragma unroll
  for (int k = 0; k < G; ++k)
    if (has[k]) dst[tid + 1024 * k] = g[k];
  if (twt_part && tid == 0) twt_part[blockIdx.x] = twt;
}

template <int G, int RB, int D>
__global__ __launch_bounds__(1024) void dense_fused_f64_ring_kernel(
    int64_t n, int64_t P, int64_t ld, int64_t rows_per_wg,
    const double* __restrict__ X, const double* __restrict__ v,
    const double* __restrict__ rowscale, double* __restrict__ slab,
    const int* __restrict__ skip_flag, double* __restrict__ twt_part,
    const double* __restrict__ addend) {
  static_assert(RB == 1 || RB == 2, "rows per ring stage");
  if (skip_flag && *skip_flag) return;  // the CG solve has already stopped
  double twt = 0.;
  constexpr int SLOT_Q = G * 1024;      // 16-byte units per row slot
  constexpr int NWAVE = 1024 / WAVE;
  extern __shared__ __attribute__((aligned(16))) unsigned char fused_smem[];
  fused_d2* ring = reinterpret_cast<fused_d2*>(fused_smem);  // [D * RB][SLOT_Q]
  double* red = reinterpret_cast<double*>(fused_smem + (size_t)D * RB * SLOT_Q * 16);
  double* rs = red + 2 * 2 * NWAVE;     // [rows_per_wg + 8]
  double* ad = rs + rows_per_wg + 8;    // [rows_per_wg + 8]
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int64_t ldp = ld / 2;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = (r0 + rows_per_wg < n) ? r0 + rows_per_wg : n;
  fused_d2 vo[G], g[G];
  bool has[G];
  fused_f64_setup<G>(tid, ldp, P, v, vo, g, has);
  int64_t qs[G];  // source unit of this thread (clamped: every lane issues)
#pragma unroll
  for (int k = 0; k < G; ++k) qs[k] = has[k] ? tid + 1024 * k : ldp - 1;
  fused_d2* __restrict__ dst =
      reinterpret_cast<fused_d2*>(slab + (int64_t)blockIdx.x * ld);
  if (r0 >= r1) {  // no rows: a zero slab
#pragma unroll
    for (int k = 0; k < G; ++k)
      if (has[k]) dst[tid + 1024 * k] = g[k];
    if (twt_part && tid == 0) twt_part[blockIdx.x] = 0.;
    return;
  }
  const int n_rows = (int)(r1 - r0);
  const int n_blk = (n_rows + RB - 1) / RB;
  for (int j = tid; j < n_blk * RB; j += 1024) {
    rs[j] = j < n_rows ? (rowscale ? rowscale[r0 + j] : 1.) : 0.;
    ad[j] = (j < n_rows && addend) ? addend[r0 + j] : 0.;
  }
  if (tid < 2 * 2 * NWAVE) red[tid] = 0.;   // (RB == 1: the second row stays 0)
  // every compiler-visible load is retired before the counted ring starts
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
  __syncthreads();
  const fused_d2* __restrict__ X2 = reinterpret_cast<const fused_d2*>(X);
  const unsigned ring_lds = (unsigned)(uintptr_t)ring;
  // Rows past the end are clamped to the last row (their scale is 0): every
  // wave issues exactly RB * G DMA instructions per block (the counted waits)
  auto issue = [&](int b, int slot) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      int64_t row = r0 + (int64_t)b * RB + i;
      if (row >= r1) row = r1 - 1;
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const unsigned dst_lds = __builtin_amdgcn_readfirstlane(
            ring_lds + (unsigned)(((slot * RB + i) * SLOT_Q + k * 1024 + wave * WAVE) * 16));
        glds16(X2 + row * ldp + qs[k], dst_lds);
      }
    }
  };
  constexpr int PER_BLOCK = RB * G;
#pragma unroll
  for (int b = 0; b < D; ++b)
    if (b < n_blk) issue(b, b);
  int slot = 0, buf = 0;
  for (int b = 0; b < n_blk; ++b) {
    if (b + D <= n_blk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PER_BLOCK) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    fused_d2 xc[RB][G];
    double sc[RB], ac[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      sc[i] = rs[b * RB + i];
      ac[i] = ad[b * RB + i];
#pragma unroll
      for (int k = 0; k < G; ++k) {
        xc[i][k] = ring[(size_t)(slot * RB + i) * SLOT_Q + k * 1024 + tid];
        if (!has[k]) xc[i][k] = fused_d2{0., 0.};
      }
    }
    // the slot's bytes are in registers: refill it D blocks ahead
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (b + D < n_blk) issue(b + D, slot);
    // row sums: the two-row exchange of the register form (RB == 1: its second
    // row is 0.0 -- the first row's additions are the same ones, in order)
    double t[2];
    wave_sum_pair(fused_row_dot2<G>(xc[0], vo),
                  RB == 2 ? fused_row_dot2<G>(xc[RB - 1], vo) : 0., t[0], t[1]);
    if (lane == 0) {
      red[(buf * 2 + 0) * NWAVE + wave] = t[0];
      if (RB == 2) red[(buf * 2 + 1) * NWAVE + wave] = t[1];
    }
    // LDS writes visible, then the barrier; the DMAs in flight are NOT waited for
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    block_sum_pair(red + buf * 2 * NWAVE, lane, t[0], t[1]);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const double wi = sc[i] * t[i];
      twt = fma(wi, t[i], twt);
      fused_row_axpy2<G>(xc[i], wi + ac[i], g);
    }
    buf ^= 1;
    slot = (slot + 1 == D) ? 0 : slot + 1;
  }
#pragma unroll
  for (int k = 0; k < G; ++k)
    if (has[k]) dst[tid + 1024 * k] = g[k];
  if (twt_part && tid == 0) twt_part[blockIdx.x] = twt;
}

static size_t fused_f64_ring_lds(int G, int RB, int D, int64_t rows_per_wg) {
  return (size_t)D * RB * G * 1024 * 16 + sizeof(double) * 2 * 2 * 16 +
         2 * sizeof(double) * (size_t)(rows_per_wg + 8);
}

static size_t fused_ring_lds(int KQ, int RB, int D, int64_t rows_per_wg) {
  return (size_t)D * RB * KQ * 1024 * 16 + sizeof(double) * 2 * RB * 16 +
         2 * sizeof(double) * (size_t)(rows_per_wg + 8);
}

// Does the single-pass operator kernel apply to this design?
bool dense_fused_applies(const bbx_design* h) {
  // one or two column groups (4 columns each) per thread; with f64 storage two
  // groups take 112 of the 128 registers
  const int64_t ld_max = 8192;
  return !(h->sparse || h->dense_ld > ld_max || h->n < 4096);
}

int launch_operator_dense_fused(bbx_design* h, const double* d_v,
                                const double* d_rowscale,
                                const TdotEpilogue& ep, double* d_out,
                                double* d_twt_part, const double* d_addend) {
  static_assert(NPART == 256, "one <t, Omega t> partial per workgroup");
  // BBX_DENSE_FUSED_RING=0 keeps the register-prefetch kernel everywhere;
  // unset: the LDS-DMA ring for f32 storage from 64 rows per workgroup on (its
  // prologue costs more than it saves on short row ranges)
  static const int ring_env =
      getenv("BBX_DENSE_FUSED_RING") ? atoi(getenv("BBX_DENSE_FUSED_RING")) : -1;
  if (!dense_fused_applies(h)) return 1;
  const int wgs = 256;
  if (h->dense_fused_wgs != wgs) {
    BBX_TRY(h->dense_fused_slab.alloc(sizeof(double) * (size_t)wgs *
                                      (size_t)h->dense_ld));
    h->dense_fused_wgs = wgs;
  }
  const int64_t rows_per_wg = (h->n + wgs - 1) / wgs;
  const bool kq1 = h->dense_ld <= 4096;
  h->n_dot += 1;
  h->n_tdot += 1;
  BBX_TRY(timer_begin(h, 0));
#define BBX_FUSED_LAUNCH(TT, KQ, RB)                                           \
  BBX_LAUNCH((dense_fused_kernel<TT, KQ, RB>), dim3(wgs), dim3(1024),  \
                     0, h->stream, h->n, h->P, h->dense_ld, rows_per_wg,       \
                     h->dense.as<TT>(), d_v, d_rowscale,                       \
                     h->dense_fused_slab.as<double>(), h->skip_flag,           \
                     d_twt_part, d_addend)
#define BBX_RING_LAUNCH(KQ, RB, D)                                             \
  do {                                                                         \
    const size_t lb = fused_ring_lds(KQ, RB, D, rows_per_wg);                  \
    /* per device, cheap next to a 1 ms kernel: set on every launch */        \
    BBX_HIP(hipFuncSetAttribute(                                               \
        reinterpret_cast<const void*>(&dense_fused_ring_kernel<KQ, RB, D>),    \
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));              \
    BBX_LAUNCH((dense_fused_ring_kernel<KQ, RB, D>), dim3(wgs),        \
                       dim3(1024), lb, h->stream, h->n, h->P, h->dense_ld,     \
                       rows_per_wg, h->dense.as<float>(), d_v, d_rowscale,     \
                       h->dense_fused_slab.as<double>(), h->skip_flag,         \
                       d_twt_part, d_addend);                                          \
  } while (0)
  const bool ring = h->dense_dtype == BBX_F32 && ring_env != 0 &&
                    (ring_env > 0 || rows_per_wg >= 64) &&
                    fused_ring_lds(2, 2, 2, rows_per_wg) <= 160 * 1024;
  // f64 storage: pairs of doubles per thread and group (see the kernels); a
  // stage of the ring is one row of <= 8192 doubles or two of <= 4096 (64 KB)
#define BBX_F64_LAUNCH(GG)                                                     \
  BBX_LAUNCH((dense_fused_f64_kernel<GG>), dim3(wgs), dim3(1024), 0,   \
                     h->stream, h->n, h->P, h->dense_ld, rows_per_wg,          \
                     h->dense.as<double>(), d_v, d_rowscale,                   \
                     h->dense_fused_slab.as<double>(), h->skip_flag,           \
                     d_twt_part, d_addend)
#define BBX_F64_RING_LAUNCH(GG, RB, D)                                         \
  do {                                                                         \
    const size_t lb = fused_f64_ring_lds(GG, RB, D, rows_per_wg);              \
    BBX_HIP(hipFuncSetAttribute(                                               \
        reinterpret_cast<const void*>(&dense_fused_f64_ring_kernel<GG, RB, D>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));              \
    BBX_LAUNCH((dense_fused_f64_ring_kernel<GG, RB, D>), dim3(wgs),    \
                       dim3(1024), lb, h->stream, h->n, h->P, h->dense_ld,     \
                       rows_per_wg, h->dense.as<double>(), d_v, d_rowscale,    \
                       h->dense_fused_slab.as<double>(), h->skip_flag,         \
                       d_twt_part, d_addend);                                  \
  } while (0)
  if (h->dense_dtype != BBX_F32) {
    const bool g2 = h->dense_ld <= 4096;
    // measured at 200k x 8k (profiles/r04_ab_dense_fused.txt): register form
    // 1.85 ms = 0.87 of peak, ring 1.91 ms = 0.84 -- with 16-byte lane loads the
    // register prefetch already streams at the f32 ring's rate, and one row per
    // stage means a barrier per row; the ring is opt-in (BBX_DENSE_FUSED_RING=22)
    const bool ring64 = ring_env > 0 &&
                        fused_f64_ring_lds(4, 1, 2, rows_per_wg) <= 160 * 1024;
    if (ring64) {
      if (g2) BBX_F64_RING_LAUNCH(2, 2, 2); else BBX_F64_RING_LAUNCH(4, 1, 2);
    } else {
      if (g2) BBX_F64_LAUNCH(2); else BBX_F64_LAUNCH(4);
    }
  } else if (ring) {
    if (kq1) BBX_RING_LAUNCH(1, 2, 2); else BBX_RING_LAUNCH(2, 2, 2);
  } else if (kq1) {
    BBX_FUSED_LAUNCH(float, 1, 2);
  } else {
    BBX_FUSED_LAUNCH(float, 2, 2);
  }
#undef BBX_FUSED_LAUNCH
#undef BBX_RING_LAUNCH
#undef BBX_F64_LAUNCH
#undef BBX_F64_RING_LAUNCH
  BBX_TRY(timer_end(h, 0));
  BBX_HIP(hipGetLastError());
  return launch_tdot_finalize_dense(h, ep, d_out,
                                    h->dense_fused_slab.as<double>(), wgs);
}

// Device copy with centring, intercept column and zero padding:
//   dst[i, 0] = 1 (intercept), dst[i, a + j] = src[i, j] - offset[j]
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void dense_ingest_kernel(
    int64_t n, int64_t p, int64_t ld, int intercept,
    const TIN* __restrict__ src, const double* __restrict__ offset,
    TOUT* __restrict__ dst) {
  const int64_t total = n * ld;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total;
       k += (int64_t)gridDim.x * 256) {
    const int64_t i = k / ld, c = k - i * ld;
    double val = 0.;
    if (intercept && c == 0) {
      val = 1.;
    } else if (c - intercept < p && c >= intercept) {
      const int64_t j = c - intercept;
      val = (double)src[i * p + j];
      if (offset) val -= offset[j];
    }
    dst[k] = (TOUT)val;
  }
}

int launch_dot_dense(bbx_design* h, const double* d_v,
                     const double* d_rowscale, double* d_t) {
  int64_t nb = (h->n + 3) / 4;
  if (nb > 1024) nb = 1024;   // 4 rows in flight per block, vector staged once
  if (nb < 1) nb = 1;
  const size_t lds = sizeof(double) * (size_t)h->dense_ld;
  if (lds > 150 * 1024)
    return fail(BBX_ERR_INVALID, "dense operator: more than 19200 columns");
  // opt-in matrix-core variant (A/B only: LABNOTES.md 3.3)
  static const bool use_mfma =
      getenv("BBX_DENSE_MFMA") && atoi(getenv("BBX_DENSE_MFMA")) == 1;
  BBX_TRY(timer_begin(h, 0));
  if (h->dense_dtype == BBX_F32 && use_mfma)
    BBX_LAUNCH(dense_dot_mfma_kernel, dim3(1024), dim3(256),
                       lds + 16 * sizeof(double), h->stream, h->n, h->P,
                       h->dense_ld, h->dense.as<float>(), d_v, d_rowscale, d_t);
  else if (h->dense_dtype == BBX_F32)
    BBX_LAUNCH(dense_dot_kernel<float>, dim3((unsigned)nb), dim3(256),
                       lds, h->stream, h->n, h->P, h->dense_ld,
                       h->dense.as<float>(), d_v, d_rowscale, d_t, h->skip_flag);
  else
    BBX_LAUNCH(dense_dot_kernel<double>, dim3((unsigned)nb), dim3(256),
                       lds, h->stream, h->n, h->P, h->dense_ld,
                       h->dense.as<double>(), d_v, d_rowscale, d_t,
                       h->skip_flag);
  BBX_TRY(timer_end(h, 0));
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

static int launch_tdot_slabs_dense(bbx_design* h, const double* d_w) {
  const int64_t ld = h->dense_ld;
  const int chunks = h->dense_chunks;
  const int64_t rows_per_chunk = (h->n + chunks - 1) / chunks;
  const dim3 grid((unsigned)((ld / 4 + 255) / 256), (unsigned)chunks);
  BBX_TRY(timer_begin(h, 1));
  if (h->dense_dtype == BBX_F32)
    BBX_LAUNCH(dense_tdot_kernel<float>, grid, dim3(256), 0, h->stream,
                       h->n, ld, rows_per_chunk, h->dense.as<float>(), d_w,
                       h->dense_slab.as<double>(), h->skip_flag);
  else
    BBX_LAUNCH(dense_tdot_kernel<double>, grid, dim3(256), 0,
                       h->stream, h->n, ld, rows_per_chunk,
                       h->dense.as<double>(), d_w, h->dense_slab.as<double>(),
                       h->skip_flag);
  BBX_TRY(timer_end(h, 1));
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_tdot_dense(bbx_design* h, const double* d_w,
                      const double* /*d_sumw_part*/, const TdotEpilogue& ep,
                      double* d_out) {
  BBX_TRY(launch_tdot_slabs_dense(h, d_w));
  // The intercept column and the centring live in the matrix itself, so the
  // common epilogue runs with intercept = 0, offset = 0, sum(w) unused.
  return launch_tdot_finalize_dense(h, ep, d_out);
}

static int create_dense_common(int64_t n, int64_t p, const void* X,
                               int in_dtype, int storage_dtype,
                               const double* col_offset, int add_intercept,
                               int device, bool from_device,
                               bbx_design** out) {
  if (!out) return fail(BBX_ERR_INVALID, "out handle pointer is NULL");
  *out = nullptr;
  if (n <= 0 || p <= 0 || !X)
    return fail(BBX_ERR_INVALID, "n and p must be positive, X not NULL");
  if ((in_dtype != BBX_F32 && in_dtype != BBX_F64) ||
      (storage_dtype != BBX_F32 && storage_dtype != BBX_F64))
    return fail(BBX_ERR_INVALID, "unknown dtype");
  bbx_design* h = new (std::nothrow) bbx_design();
  if (h) design_register(h);
  if (!h) return fail(BBX_ERR_INVALID, "out of host memory");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    design_unregister(h);
    delete h;
    return fail(BBX_ERR_NODEVICE,
                "no HIP device visible (libbbx has no CPU fallback)");
  }
  if (device < 0 || device >= count) {
    design_unregister(h);
    delete h;
    return fail(BBX_ERR_INVALID, "device index out of range");
  }
  auto body = [&]() -> int {
    SetupTurn turn;   // (ranks sharing a GPU: one device set-up at a time)
    BBX_HIP(hipSetDevice(device));
    h->device = device;
    BBX_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->n = n;
    h->p = p;
    h->intercept = add_intercept ? 1 : 0;
    h->P = p + h->intercept;
    h->nnz = n * h->P;
    h->sparse = false;
    h->format = 0;
    h->centred = col_offset != nullptr;
    h->dense_dtype = storage_dtype;
    h->dense_ld = (h->P + 7) / 8 * 8;
    h->dense_chunks = DENSE_ROW_CHUNKS;
    if (h->n < 4 * DENSE_ROW_CHUNKS) h->dense_chunks = 1;
    BBX_TRY(design_alloc_work(h));
    const size_t el_out = storage_dtype == BBX_F32 ? 4 : 8;
    const size_t el_in = in_dtype == BBX_F32 ? 4 : 8;
    // DENSE_PAD_ROWS zero rows behind the matrix: the batched products' register
    // rings read ahead of the rows and columns they use (dense_batch.hip)
    BBX_TRY(h->dense.alloc(el_out * (size_t)(n + DENSE_PAD_ROWS) *
                           (size_t)h->dense_ld));
    BBX_HIP(hipMemset(static_cast<char*>(h->dense.ptr) +
                          el_out * (size_t)n * (size_t)h->dense_ld,
                      0, el_out * (size_t)DENSE_PAD_ROWS * (size_t)h->dense_ld));
    BBX_TRY(h->dense_slab.alloc(sizeof(double) * (size_t)h->dense_chunks *
                                (size_t)h->dense_ld));
    // zeros for the epilogue's offset (length P) and sum(w) partials
    BBX_TRY(h->offset.alloc(sizeof(double) * (size_t)h->P));
    BBX_HIP(hipMemset(h->offset.ptr, 0, sizeof(double) * (size_t)h->P));
    DevMem d_src, d_off;
    const void* src = X;
    const double* off = col_offset;
    if (!from_device) {
      BBX_TRY(d_src.alloc(el_in * (size_t)n * (size_t)p));
      BBX_HIP(hipMemcpy(d_src.ptr, X, el_in * (size_t)n * (size_t)p,
                        hipMemcpyHostToDevice));
      src = d_src.ptr;
      if (col_offset) {
        BBX_TRY(d_off.alloc(sizeof(double) * (size_t)p));
        BBX_HIP(hipMemcpy(d_off.ptr, col_offset, sizeof(double) * (size_t)p,
                          hipMemcpyHostToDevice));
        off = d_off.as<double>();
      }
    }
    // h->stream is non-blocking: wait for the null-stream memsets/copies above
    // (and for whatever produced a device-resident X) before reading them
    BBX_HIP(hipDeviceSynchronize());
    const dim3 grid(4096), block(256);
    if (in_dtype == BBX_F32 && storage_dtype == BBX_F32)
      BBX_LAUNCH((dense_ingest_kernel<float, float>), grid, block, 0,
                         h->stream, n, p, h->dense_ld, h->intercept,
                         (const float*)src, off, h->dense.as<float>());
    else if (in_dtype == BBX_F32)
      BBX_LAUNCH((dense_ingest_kernel<float, double>), grid, block, 0,
                         h->stream, n, p, h->dense_ld, h->intercept,
                         (const float*)src, off, h->dense.as<double>());
    else if (storage_dtype == BBX_F32)
      BBX_LAUNCH((dense_ingest_kernel<double, float>), grid, block, 0,
                         h->stream, n, p, h->dense_ld, h->intercept,
                         (const double*)src, off, h->dense.as<float>());
    else
      BBX_LAUNCH((dense_ingest_kernel<double, double>), grid, block, 0,
                         h->stream, n, p, h->dense_ld, h->intercept,
                         (const double*)src, off, h->dense.as<double>());
    BBX_HIP(hipGetLastError());
    BBX_HIP(hipStreamSynchronize(h->stream));
    BBX_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&dense_dot_kernel<float>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    BBX_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&dense_dot_kernel<double>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    BBX_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(&dense_dot_mfma_kernel),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return BBX_OK;
  };
  int st = no_throw(body);
  if (st < 0) {
    bbx_design_destroy(h);
    return st;
  }
  *out = h;
  return BBX_OK;
}

}  // namespace bbx

extern "C" {

int bbx_design_create_dense(int64_t n, int64_t p, const void* X, int in_dtype,
                            int storage_dtype, const double* col_offset,
                            int add_intercept, int device, bbx_design** out) {
  return bbx::create_dense_common(n, p, X, in_dtype, storage_dtype, col_offset,
                                  add_intercept, device, false, out);
}

int bbx_design_create_dense_dev(int64_t n, int64_t p, const void* d_X,
                                int in_dtype, int storage_dtype,
                                const double* d_col_offset, int add_intercept,
                                int device, bbx_design** out) {
  return bbx::create_dense_common(n, p, d_X, in_dtype, storage_dtype,
                                  d_col_offset, add_intercept, device, true,
                                  out);
}

}  // extern "C"
