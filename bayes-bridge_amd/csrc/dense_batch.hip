// Dense design, batched chains: the two products of the CG operator for K <= 16
// right-hand sides at once, on the matrix cores.
//
// A GEMV cannot feed v_mfma_f64_16x16x4_f64 (one right-hand side: 15 of the 16
// columns of B are padding; measured 1.7x SLOWER than the vector ALUs,
// DESIGN.md 3.3).  K chains that share the pass over X are the shape that can:
//   T[n x K] = X[n x P] V[P x K]          (dense_matrix.py:42, K at a time)
//   G[P x K] = X^T[P x n] W[n x K]        (dense_matrix.py:52, K at a time)
// with the chains in the 16 columns of B / D.  The single-chain path fuses the
// two products into one pass (dense.hip) because its slice of v and of the
// result fits a thread's registers; for K chains that state is 2 P K doubles
// per workgroup (512 KB at K = 4: a CU's whole register file), so the batch
// reads X twice per operator application -- for all its chains.
//
// Both kernels stream X the same way: a wavefront owns a 64 row x 64 column
// stage (16 KB of f32), filled by 16 LDS-DMA instructions (global_load_lds_
// dwordx4: 4 rows x 256 contiguous bytes each, no register destination) into
// its PRIVATE slice of LDS, two stages deep; only the issuing wave's counted
// vmcnt orders the data, there is no workgroup barrier in the loops.  The
// 16-byte quads of a row are stored XOR-swizzled with the row index
// (slot = quad ^ (row & 15)): the ds_read_b128 of the A operand (16 rows x the
// same quad per lane group) then touches 16 distinct 16-byte slots of the bank
// row -- conflict free (MI355X_MICROARCH.md "LDS"; the DMA's lanes simply read
// their quad from the permuted global address, still one 256-byte segment).
// A stage feeds 64 MFMAs; the B operand (8 KB per stage: V or W, L2 resident)
// comes straight from global memory into registers, one stage ahead.
// Four waves per CU (one per SIMD), 128 KB of LDS.
//
// Operand maps (cdna_hip_programming.md "f64 MFMA"): A[l & 15][l >> 4],
// B[l >> 4][l & 15], D col = l & 15, row = (l >> 4) + 4 reg.
// Every column of D is computed from its own column of B only: a chain's
// numbers do not depend on the other chains of the batch (bit for bit).
#include "common.hpp"

// -DDK_ABLATE=1: no DMA (stale LDS), =2: no MFMAs -- timing only, wrong results
#ifndef DK_ABLATE
#define DK_ABLATE 0
#endif

namespace bbx {

typedef double dk_d4 __attribute__((ext_vector_type(4)));

// Stage geometry, measured at 200k x 8k, K = 16 (profiles/r03_dense_batch.txt;
// 128 KB of LDS per CU either way, two stages per wave):
//   X V   : 64 rows x 4 waves 1.57 ms, 32 rows x 8 waves 1.72 ms
//   X^T W : 64 rows x 4 waves 1.37 ms, 32 rows x 8 waves 2.29 ms
// (before the stage's A fragments were all read up front and the operand loads
// lost their conditions the same shapes took 3.5 / 2.1 and 2.1 / 2.9 ms)
#ifndef DK_DOT_ROWS
#define DK_DOT_ROWS 64
#endif
#ifndef DK_TDOT_ROWS
#define DK_TDOT_ROWS 64
#endif
constexpr int DK_COLS = 64;                 // columns of a stage (256 B of f32)
constexpr int DK_STAGES = 2;
constexpr int DK_KS = DENSE_BATCH_STRIDE;   // interleave stride: the 16 columns of B
template <int ROWS>
struct DkGeom {
  static constexpr int rows = ROWS;             // rows of a stage
  static constexpr int nt = ROWS / 16;          // 16-row tiles of a stage
  static constexpr int stage_bytes = ROWS * DK_COLS * 4;
  static constexpr int waves = 256 / ROWS;      // per workgroup (and CU)
  static constexpr int lds_bytes = waves * DK_STAGES * stage_bytes;  // 128 KB
};
using DkDot = DkGeom<DK_DOT_ROWS>;
using DkTdot = DkGeom<DK_TDOT_ROWS>;
constexpr int DK_DOT_WGS = 256;             // one workgroup per CU
constexpr int DK_TDOT_CHUNKS = 8;           // row chunks of the transposed product

__device__ __forceinline__ void dk_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// ROWS / 4 DMA instructions: rows [row0, row0 + ROWS) x columns [col0, col0 + 64) of
// the row-major f32 matrix into the wave's stage.  Rows / quads past the end
// are clamped to the last valid ones (finite values that only ever meet a zero
// B operand), so that every wave issues the same number of DMAs.
template <int ROWS>
__device__ __forceinline__ void dk_issue_stage(const float* __restrict__ X,
                                               int64_t n /* row clamp */,
                                               int64_t ld, int64_t row0,
                                               int64_t col0,
                                               unsigned stage_lds, int lane) {
  const int64_t ldq = ld / 4;
#pragma unroll
  for (int d = 0; d < ROWS / 4; ++d) {
    const int r_in = 4 * d + (lane >> 4);
    int64_t row = row0 + r_in;
    if (row >= n) row = n - 1;
    int64_t quad = col0 / 4 + ((lane & 15) ^ (r_in & 15));
    if (quad >= ldq) quad = ldq - 1;
    const unsigned dst =
        __builtin_amdgcn_readfirstlane(stage_lds + (unsigned)(d * 1024));
    if (DK_ABLATE != 1)
      dk_glds16(reinterpret_cast<const float4*>(X) + row * ldq + quad, dst);
  }
}

// One group of NT 16-row tiles of T = X V (NT = DkDot::nt: the B operand of a stage
// is used NT times, consecutive MFMAs go to NT accumulators; NT = 1, the
// remainder of a wave's range: one accumulator per k-slot, added at the end).
// Straight-line per stage: no branch sits between two MFMAs.
template <int NT>
__device__ __forceinline__ void dk_dot_group(int K, 
    int64_t n, int64_t P, int64_t ld, const float* __restrict__ X,
    const double* __restrict__ v, const ChainPtrs& rowscale,
    const ChainOut& out, int out_stride, int64_t tile, unsigned my_lds,
    const unsigned char* my_stage, int lane, double& twt) {
  const int i = lane & 15, k = lane >> 4;
  const int64_t row0 = tile * 16;
  int64_t row_lim = (tile + NT) * 16;  // rows past the group: its last row again
  if (row_lim > n) row_lim = n;
  const int n_stage = (int)((ld + DK_COLS - 1) / DK_COLS);
  constexpr int NACC = NT == 1 ? 4 : NT;  // NT == 1: one accumulator per m
  dk_d4 D[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) D[a] = dk_d4{0., 0., 0., 0.};
  double bn[16], bc[16];
  // B operand: V is [ld + 64][16], zero padded in both directions (rows past
  // P, columns past the batch's chains), so the loads carry no condition
  auto load_b = [&](int s, double (&b)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int64_t col = (int64_t)s * DK_COLS + 16 * c + 4 * k + m;
        b[4 * c + m] = v[col * DK_KS + i];
      }
  };
  dk_issue_stage<DkDot::rows>(X, row_lim, ld, row0, 0, my_lds, lane);
  load_b(0, bn);
  for (int s = 0; s < n_stage; ++s) {
    const int slot = s & 1;
    // Stage s and its B operands have landed.  (The B loads are visible to the
    // compiler, which would wait for them with vmcnt(0) anyway -- it does not
    // count the DMAs -- so the wait comes BEFORE the next stage's requests are
    // issued: exactly one stage is in flight during the MFMAs.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 16; ++u) bc[u] = bn[u];
    if (s + 1 < n_stage) {
      dk_issue_stage<DkDot::rows>(X, row_lim, ld, row0, (int64_t)(s + 1) * DK_COLS,
                     my_lds + (unsigned)((slot ^ 1) * DkDot::stage_bytes), lane);
      load_b(s + 1, bn);
    }
    const float4* st4 =
        reinterpret_cast<const float4*>(my_stage + slot * DkDot::stage_bytes);
    // all A fragments of the stage first (one exposed LDS round trip per
    // stage instead of one per 16-column step: with one or two waves per SIMD
    // nothing else hides it), then the MFMAs back to back
    float4 xa[4][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int rt = 0; rt < NT; ++rt)
        xa[c][rt] = st4[(16 * rt + i) * 16 + ((4 * c + k) ^ i)];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double a[NT][4];
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) {
        a[rt][0] = (double)xa[c][rt].x;
        a[rt][1] = (double)xa[c][rt].y;
        a[rt][2] = (double)xa[c][rt].z;
        a[rt][3] = (double)xa[c][rt].w;
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
          constexpr bool split = NT == 1;
          dk_d4& acc = D[split ? m : rt];
          if (DK_ABLATE == 2) acc[0] += a[rt][m] * bc[4 * c + m];
          else
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt][m], bc[4 * c + m], acc,
                                                     0, 0, 0);
        }
    }
    // every lane has read its part of the stage before this slot is refilled
    // (next iteration, same wave: program order plus the LDS counter)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (NT == 1) D[0] = (D[0] + D[1]) + (D[2] + D[3]);
  // D[rt][reg] = t[row0 + 16 rt + k + 4 reg][chain i]
  if (i < K) {
    const double* rs = rowscale.p[i];
    double* o = out.p[i];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t row = row0 + 16 * rt + k + 4 * reg;
        if (row < n) {
          const double t = D[rt][reg];
          double w = t;
          if (rs) w *= rs[row];
          o[row * out_stride] = w;
          twt = fma(w, t, twt);
        }
      }
  }
}

// T = X V for K interleaved right-hand sides: out.p[c][row * out_stride] =
// rowscale_c[row] * <X[row, :], v_c>, and per workgroup and chain the partials
// of sum_i rowscale_c,i t_c,i^2 (twt_part[c * NPART + blockIdx.x]).
__global__ __launch_bounds__(DkDot::waves * WAVE) void dense_dot_k_kernel(
    int K, int64_t n, int64_t P, int64_t ld, const float* __restrict__ X,
    const double* __restrict__ v, ChainPtrs rowscale, ChainOut out,
    int out_stride, double* __restrict__ twt_part,
    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char dk_smem[];
  __shared__ double s_twt[DkDot::waves][16];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const unsigned my_lds = (unsigned)(uintptr_t)dk_smem +
                          (unsigned)(wave * DK_STAGES * DkDot::stage_bytes);
  const unsigned char* my_stage = dk_smem + wave * DK_STAGES * DkDot::stage_bytes;
  // contiguous ranges of 16-row tiles per wave, as even as 16-row tiles allow
  const int64_t n_tile = (n + 15) / 16;
  const int64_t n_wave = (int64_t)gridDim.x * DkDot::waves;
  const int64_t gw = (int64_t)blockIdx.x * DkDot::waves + wave;
  const int64_t base = n_tile / n_wave, extra = n_tile % n_wave;
  const int64_t t0 = gw * base + (gw < extra ? gw : extra);
  const int64_t t1 = t0 + base + (gw < extra ? 1 : 0);
  double twt = 0.;  // this lane's part of <t_c, Omega_c t_c>, c = lane & 15
  int64_t tile = t0;
  for (; tile + DkDot::nt <= t1; tile += DkDot::nt)
    dk_dot_group<DkDot::nt>(K, n, P, ld, X, v, rowscale, out, out_stride, tile, my_lds,
                       my_stage, lane, twt);
  for (; tile < t1; ++tile)
    dk_dot_group<1>(K, n, P, ld, X, v, rowscale, out, out_stride, tile, my_lds,
                       my_stage, lane, twt);
  if (twt_part) {
    // lanes i, i + 16, i + 32, i + 48 hold chain i's parts: fixed order
    // (t_l + t_l+16) + (t_l+32 + t_l+48) in lanes 0-15
    double a = twt + __shfl_xor(twt, 16);
    a = a + __shfl_xor(a, 32);
    if (lane < 16) s_twt[wave][lane] = a;
    __syncthreads();
    if (tid < K) {
      double tot = 0.;
      for (int wv = 0; wv < DkDot::waves; ++wv) tot += s_twt[wv][tid];
      twt_part[tid * NPART + blockIdx.x] = tot;
    }
  }
}

// Slabs of G = X^T W for K interleaved right-hand sides: a wave owns 64
// columns of X and one of DK_TDOT_CHUNKS row ranges;
// slab[(chunk * ld + col) * K + c] = sum over the chunk's rows.
__global__ __launch_bounds__(DkTdot::waves * WAVE) void dense_tdot_k_kernel(
    int K, int64_t n, int64_t ld, int64_t rows_per_chunk, int n_colblk,
    const float* __restrict__ X, const double* __restrict__ w,
    double* __restrict__ slab, const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char dk_smem[];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int i = lane & 15, k = lane >> 4;
  const unsigned my_lds = (unsigned)(uintptr_t)dk_smem +
                          (unsigned)(wave * DK_STAGES * DkTdot::stage_bytes);
  const unsigned char* my_stage = dk_smem + wave * DK_STAGES * DkTdot::stage_bytes;
  // consecutive waves take adjacent column blocks of the same row chunk
  const int64_t gw = (int64_t)blockIdx.x * DkTdot::waves + wave;
  const int chunk = (int)(gw / n_colblk);
  const int colblk = (int)(gw - (int64_t)chunk * n_colblk);
  if (chunk >= DK_TDOT_CHUNKS) return;
  const int64_t col0 = (int64_t)colblk * DK_COLS;
  const int64_t r_begin = (int64_t)chunk * rows_per_chunk;
  int64_t r_end = r_begin + rows_per_chunk;
  if (r_end > n) r_end = n;
  const int n_stage =
      r_end > r_begin ? (int)((r_end - r_begin + DkTdot::rows - 1) / DkTdot::rows) : 0;
  dk_d4 D[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) D[ct] = dk_d4{0., 0., 0., 0.};
  double bn[16], bc[16];
  auto load_b = [&](int s, double (&b)[16]) {
#pragma unroll
    for (int r = 0; r < DkTdot::rows / 4; ++r) {
      // W is [n + 64][16], zero padded (rows past n, columns past the batch's
      // chains); a chunk is a whole number of stages, so no stage straddles two
      const int64_t row = r_begin + (int64_t)s * DkTdot::rows + 4 * r + k;
      b[r] = w[row * DK_KS + i];
    }
  };
  if (n_stage > 0) {
    dk_issue_stage<DkTdot::rows>(X, n, ld, r_begin, col0, my_lds, lane);
    load_b(0, bn);
  }
  for (int s = 0; s < n_stage; ++s) {
    const int slot = s & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // see dense_dot_k_kernel
#pragma unroll
    for (int u = 0; u < 16; ++u) bc[u] = bn[u];
    if (s + 1 < n_stage) {
      dk_issue_stage<DkTdot::rows>(X, n, ld, r_begin + (int64_t)(s + 1) * DkTdot::rows, col0,
                     my_lds + (unsigned)((slot ^ 1) * DkTdot::stage_bytes), lane);
      load_b(s + 1, bn);
    }
    const float* st1 =
        reinterpret_cast<const float*>(my_stage + slot * DkTdot::stage_bytes);
    // A[i][k] of (row group r, column tile ct) = X[4 r + k][16 ct + i]; the
    // whole stage's fragments first, then the MFMAs back to back
    float xa[DkTdot::rows / 4][4];
#pragma unroll
    for (int r = 0; r < DkTdot::rows / 4; ++r) {
      const int row = 4 * r + k;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        xa[r][ct] = st1[(row * 16 + ((4 * ct + (i >> 2)) ^ (row & 15))) * 4 +
                        (i & 3)];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < DkTdot::rows / 4; ++r) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        if (DK_ABLATE == 2) D[ct][0] += (double)xa[r][ct] * bc[r];
        else
        D[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xa[r][ct], bc[r],
                                                     D[ct], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // D[ct][reg] = G[col0 + 16 ct + k + 4 reg][chain i] over this chunk's rows
  if (i < K) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t col = col0 + 16 * ct + k + 4 * reg;
        if (col < ld) slab[((int64_t)chunk * ld + col) * DK_KS + i] = D[ct][reg];
      }
  }
}

bool dense_batch_applies(const bbx_design* h) {
  return !h->sparse && h->dense_dtype == BBX_F32 && h->dense_ld % 8 == 0;
}

static int dk_set_attr() {
  BBX_HIP(hipFuncSetAttribute(
      reinterpret_cast<const void*>(&dense_dot_k_kernel),
      hipFuncAttributeMaxDynamicSharedMemorySize, DkDot::lds_bytes));
  BBX_HIP(hipFuncSetAttribute(
      reinterpret_cast<const void*>(&dense_tdot_k_kernel),
      hipFuncAttributeMaxDynamicSharedMemorySize, DkTdot::lds_bytes));
  return BBX_OK;
}

int launch_dot_dense_k(bbx_design* h, int K, const double* d_v,
                       const TiledBatchArgs& ba, double* d_twt_part) {
  if (!dense_batch_applies(h))
    return fail(BBX_ERR_STATE, "batched dense products need f32 storage");
  BBX_TRY(dk_set_attr());
  h->n_dot += 1;
  BBX_TRY(timer_begin(h, 0));
  hipLaunchKernelGGL(dense_dot_k_kernel, dim3(DK_DOT_WGS),
                     dim3(DkDot::waves * WAVE), DkDot::lds_bytes, h->stream, K,
                     h->n, h->P, h->dense_ld, h->dense.as<float>(), d_v,
                     ba.rowscale, ba.out, ba.out_stride, d_twt_part,
                     h->skip_flag);
  BBX_HIP(hipGetLastError());
  return timer_end(h, 0);
}

int launch_tdot_dense_k(bbx_design* h, int K, const double* d_w,
                        const double** slab, int* G) {
  if (!dense_batch_applies(h))
    return fail(BBX_ERR_STATE, "batched dense products need f32 storage");
  BBX_TRY(dk_set_attr());
  const size_t need = sizeof(double) * (size_t)DK_TDOT_CHUNKS *
                      (size_t)h->dense_ld * (size_t)DK_KS;
  if (h->dense_batch_slab.bytes < need) {
    BBX_TRY(h->dense_batch_slab.alloc(need));
    // columns past the batch's chains are never written: keep them zero
    BBX_HIP(hipMemsetAsync(h->dense_batch_slab.ptr, 0, need, h->stream));
  }
  const int n_colblk = (int)((h->dense_ld + DK_COLS - 1) / DK_COLS);
  const int64_t rows_per_chunk =
      ((h->n + DK_TDOT_CHUNKS - 1) / DK_TDOT_CHUNKS + DkTdot::rows - 1) /
      DkTdot::rows * DkTdot::rows;
  const int n_wave = n_colblk * DK_TDOT_CHUNKS;
  const unsigned grid = (unsigned)((n_wave + DkTdot::waves - 1) / DkTdot::waves);
  h->n_tdot += 1;
  BBX_TRY(timer_begin(h, 1));
  hipLaunchKernelGGL(dense_tdot_k_kernel, dim3(grid),
                     dim3(DkTdot::waves * WAVE), DkTdot::lds_bytes, h->stream,
                     K, h->n, h->dense_ld, rows_per_chunk, n_colblk,
                     h->dense.as<float>(), d_w,
                     h->dense_batch_slab.as<double>(), h->skip_flag);
  BBX_HIP(hipGetLastError());
  BBX_TRY(timer_end(h, 1));
  *slab = h->dense_batch_slab.as<double>();
  *G = DK_TDOT_CHUNKS;
  return BBX_OK;
}

int dense_batch_bytes(const bbx_design* h, int K, int64_t* dot_bytes,
                      int64_t* tdot_bytes) {
  // the matrix once + the 16-column operands (padding columns are read too)
  // + what the chains' columns write
  const int64_t mat = h->n * h->dense_ld * 4;
  *dot_bytes = mat + 8 * (int64_t)DK_KS * h->dense_ld + 8 * (int64_t)K * h->n;
  *tdot_bytes = mat + 8 * (int64_t)DK_KS * h->n +
                8 * (int64_t)K * DK_TDOT_CHUNKS * h->dense_ld;
  return BBX_OK;
}

}  // namespace bbx
