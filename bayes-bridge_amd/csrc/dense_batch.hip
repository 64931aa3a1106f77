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

// ---------------------------------------------------------------------------
// Direct forms: the matrix goes from HBM straight into the A-operand registers.
//
// The LDS stages above keep ONE stage in flight per wave while the previous one
// feeds the matrix cores (a second would need 256 KB), and every stage pays an
// LDS round trip before its first MFMA: 1.57 / 1.37 ms per product where the
// HBM pass is 0.9-1.0 ms and the MFMAs 0.66 ms.  A wave's registers (512 per
// lane at one wave per SIMD) hold three times what its LDS share does, and one
// orientation needs no staging at all:
//
//   G = M^T B for a row-major M: lane (i, k) loads 16 bytes M[r + k][c0 + 4 i
//   .. + 3].  A wave's load is 4 rows x 256 contiguous bytes (the DMA's shape)
//   and already IS the A operand of four MFMAs -- MFMA e takes element e of
//   every lane, i.e. the column set {c0 + 4 i + e}; which columns share a tile
//   is bookkeeping at the store.  B[k][chain] = one 512-byte load per four
//   rows, used by all 16 MFMAs of the wave's 256 columns.
//
// X^T W is that product with M = X.  X V contracts along X's CONTIGUOUS
// direction: its A operand would be 16 rows x 64 bytes per load, and at 13 row
// tiles per wave every 128-byte line is fetched twice, a slot apart (2.3 ms per
// pass, the same with the MFMAs removed; scripts/probes/row_frag_stream.hip
// shows the shape streaming well only while few lines are open).  So a batch
// keeps a TRANSPOSED copy of the matrix (dense_xt: P x n, built on first use,
// 6.4 GB at 200k x 8k -- the batch's working set is then 13 of the 288 GB) and
// computes X V = (X^T)^T V with the same sweep: both products read 4-row x
// 256-byte pieces.
//
// The sweep keeps a ring of D slots in registers (a slot = 4 rows: four A loads
// and one B load, 16 MFMAs), issued through inline asm and retired with counted
// waits (the idiom of spmv_tiled.hip): D - 1 slots -- 44 KB per wave -- are in
// flight while one is consumed.  Reads may run past a wave's rows or the end of
// the matrix (the allocations are padded with zero rows); what they fetch meets
// a zero B operand or a column that is never stored.
#ifndef DK_DIRECT
#define DK_DIRECT 1
#endif
#ifndef DKT_D
#define DKT_D 12     // ring depth (slots of 4 rows)
#endif

typedef float dk_f4 __attribute__((ext_vector_type(4)));

template <int IMM>
__device__ __forceinline__ void dk_ld_x4(dk_f4& dst, unsigned voff,
                                         const void* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt"
               : "=v"(dst)
               : "v"(voff), "s"(sbase), "n"(IMM)
               : "memory");
}
__device__ __forceinline__ void dk_ld_d(double& dst, unsigned voff,
                                        const void* sbase) {
  asm volatile("global_load_dwordx2 %0, %1, %2"
               : "=v"(dst)
               : "v"(voff), "s"(sbase)
               : "memory");
}

// The MFMA of the direct kernels: f32 -> f64 conversion of the A operand, the
// wait states between a VALU result and the matrix core's read of it, and the
// MFMA with its accumulator pinned to the accumulation half of the register
// file, as ONE asm statement.
//  * Through the builtin the compiler carries the accumulators across the loop
//    boundary in architectural VGPRs -- which the asm loads' ring already
//    fills -- and copies every one of them in and out once per period
//    (v_accvgpr_write/read x 8 per tile, each waiting for its MFMA; seen in
//    the .s).
//  * An MFMA written in asm is invisible to the compiler's hazard recogniser:
//    with the conversion left outside, `v_cvt_f64_f32 v[a:b], ..` directly in
//    front of `v_mfma .., v[a:b], ..` read the register's OLD contents (there
//    is no interlock; every column but one of the first version was wrong).
//    The s_nop below is that spacing; it also covers a B operand produced by
//    the VALU just before the statement.  The rest holds by construction: an
//    accumulator is touched again only after >= 3 other MFMAs (the statements
//    are volatile and keep their source order), and the epilogue's reads are
//    preceded by explicit s_nops.
//  The ~12 cycles of issue per statement sit in the shadow of the previous
//  MFMA's 64.
__device__ __forceinline__ void dk_mfma(dk_d4& acc, float x, double b) {
  double t;
  asm volatile(
      "v_cvt_f64_f32 %1, %2\n\ts_nop 3\n\t"
      "v_mfma_f64_16x16x4_f64 %0, %1, %3, %0"
      : "+a"(acc), "=&v"(t)
      : "v"(x), "v"(b));
}

constexpr int DKD_WAVES = 4;        // one per SIMD: 512 registers per lane
constexpr int DKD_C = 4;            // 64-column units a sweep carries
constexpr int DKD_IMG = DKD_WAVES * DKD_C * 16 * WAVE * 8;  // 128 KB of LDS

// One wave's sweep: acc[c][e] += sum over rows [r_begin, r_begin + 4 n_slot) of
// M[r][col0 + 64 c + 4 i' + e] * B[r][chain], i' the MFMA's row index.
// m_rows = rows of M that exist including its padding (addresses are clamped
// for waves without work only; the ring's read-ahead relies on the padding).
template <int C, int D>
__device__ __forceinline__ void dkd_sweep(const float* __restrict__ M,
                                          int64_t ldm, int64_t col0,
                                          int64_t r_begin, int n_slot,
                                          const double* __restrict__ Bop,
                                          int lane, dk_d4 (&acc)[DKD_C][4]) {
  static_assert(C >= 1 && C <= DKD_C, "units per sweep");
  constexpr int LPS = C + 1;
  constexpr int WAITC = (D - 1) * LPS;
  static_assert(WAITC < 64, "vmcnt is a 6-bit field");
  const int i = lane & 15, k = lane >> 4;
  const int n_period = (n_slot + D - 1) / D;
  const unsigned voff_a = (unsigned)(((int64_t)k * ldm + 4 * i) * 4);
  const unsigned voff_b = (unsigned)((k * DK_KS + i) * 8);
  // (a wave without rows primes its ring on the first rows and computes nothing)
  const int64_t r_ring = n_slot > 0 ? r_begin : 0;
  const float* sa = M + r_ring * ldm + col0;     // wave-uniform, 4 rows per slot
  const double* sb = Bop + r_ring * DK_KS;
  dk_f4 xa[D][C];
  double bw[D];
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing of the compiler's in the queue
#define DKD_ISSUE(KK)                                                         \
  do {                                                                        \
    dk_ld_x4<0>(xa[KK][0], voff_a, sa);                                       \
    if (C > 1) dk_ld_x4<256>(xa[KK][C > 1 ? 1 : 0], voff_a, sa);              \
    if (C > 2) dk_ld_x4<512>(xa[KK][C > 2 ? 2 : 0], voff_a, sa);              \
    if (C > 3) dk_ld_x4<768>(xa[KK][C > 3 ? 3 : 0], voff_a, sa);              \
    dk_ld_d(bw[KK], voff_b, sb);                                              \
    sa += 4 * ldm;                                                            \
    sb += 4 * DK_KS;                                                          \
  } while (0)
#define DKD_TIE(KK)                                                           \
  do {                                                                        \
    _Pragma("unroll") for (int c = 0; c < C; ++c)                             \
        asm volatile("" : "+v"(xa[KK][c]));                                   \
    asm volatile("" : "+v"(bw[KK]));                                          \
  } while (0)
#pragma unroll
  for (int kk = 0; kk < D; ++kk) DKD_ISSUE(kk);
  for (int p = 0; p < n_period; ++p) {
#pragma unroll
    for (int kk = 0; kk < D; ++kk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITC) : "memory");
      DKD_TIE(kk);
      // a slot past this wave's rows belongs to the next wave (or is
      // padding): it meets a zero B operand, no branch between the MFMAs
      const double b_ = (p * D + kk < n_slot) ? bw[kk] : 0.;
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (DK_ABLATE == 2) acc[c][e][0] += (double)xa[kk][c][e] * b_;
          else dk_mfma(acc[c][e], xa[kk][c][e], b_);
        }
      DKD_ISSUE(kk);
    }
  }
  // the D slots issued past the end: landed before their registers are free
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int kk = 0; kk < D; ++kk) DKD_TIE(kk);
#undef DKD_ISSUE
#undef DKD_TIE
  // 18 wait states between the last MFMA and a read of its result
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0][0]));
#pragma unroll
  for (int c = 0; c < DKD_C; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e > 0) asm volatile("" : "+a"(acc[c][e]));
}

// The four waves' accumulators through LDS, added in the fixed order
// (w0 + w1) + (w2 + w3); wave c' returns sub-block c' in g[e][reg]:
// g[e][reg] = G[col0 + 64 c' + 4 ((lane >> 4) + 4 reg) + e][chain lane & 15].
__device__ __forceinline__ void dkd_fold(const dk_d4 (&acc)[DKD_C][4],
                                         double* img, int wave, int lane,
                                         double (&g)[4][4]) {
#pragma unroll
  for (int c = 0; c < DKD_C; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        img[((((wave * DKD_C + c) * 4 + e) * 4 + reg) << 6) + lane] = acc[c][e][reg];
  __syncthreads();
  constexpr int WS = DKD_C * 16 * WAVE;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int o = (((wave * 4 + e) * 4 + reg) << 6) + lane;
      g[e][reg] = (img[o] + img[WS + o]) + (img[2 * WS + o] + img[3 * WS + o]);
    }
}

// X^T W: a workgroup = (block of 256 columns, one of DK_TDOT_CHUNKS row chunks);
// its four waves take quarters of the chunk's rows.  blockIdx.x % 8 = the
// chunk: under round-robin placement the workgroups of an XCD share a row
// chunk, and its slice of W streams through that XCD's L2 once.
__global__ __launch_bounds__(DKD_WAVES * WAVE) void dense_tdot_kd_kernel(
    int K, int64_t n, int64_t ld, int64_t rows_per_wave,
    const float* __restrict__ X, const double* __restrict__ w,
    double* __restrict__ slab, const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char dk_smem[];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int chunk = (int)(blockIdx.x % DK_TDOT_CHUNKS);
  const int colblk = (int)(blockIdx.x / DK_TDOT_CHUNKS);
  const int64_t col0 = (int64_t)colblk * (64 * DKD_C);
  const int64_t r_begin =
      ((int64_t)chunk * DKD_WAVES + wave) * rows_per_wave;  // multiple of 4
  int64_t r_end = r_begin + rows_per_wave;
  if (r_end > n) r_end = n;
  const int n_slot = r_end > r_begin ? (int)((r_end - r_begin + 3) / 4) : 0;
  dk_d4 acc[DKD_C][4];
#pragma unroll
  for (int c = 0; c < DKD_C; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[c][e] = dk_d4{0., 0., 0., 0.};
  dkd_sweep<DKD_C, DKT_D>(X, ld, col0, r_begin, n_slot, w, lane, acc);
  double g[4][4];
  dkd_fold(acc, reinterpret_cast<double*>(dk_smem), wave, lane, g);
  const int i = lane & 15, k = lane >> 4;
  if (i < K) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t col = col0 + 64 * wave + 4 * (k + 4 * reg) + e;
        if (col < ld) slab[((int64_t)chunk * ld + col) * DK_KS + i] = g[e][reg];
      }
  }
}

// X V through the transposed copy: T = (X^T)^T V, XT row-major [P + pad][ldn].
// 256 persistent workgroups; workgroup b owns the 64-row units [u0, u1) of T
// (12 or 13 of them at 200 000 rows) and sweeps all P rows of XT once per four
// units, its waves taking quarters of P; the tail of fewer than four units is
// a narrower sweep.  Epilogue per unit: rowscale, the store, <t, Omega t>.
template <int C>
__device__ __forceinline__ void dkd_dot_units(
    int K, int64_t n, int64_t P, int64_t ldn, const float* __restrict__ XT,
    const double* __restrict__ v, const ChainPtrs& rowscale, const ChainOut& out,
    int out_stride, int64_t unit0, double* img, int wave, int lane, double& twt) {
  const int64_t rows_per_wave = ((P + DKD_WAVES - 1) / DKD_WAVES + 3) / 4 * 4;
  const int64_t r_begin = (int64_t)wave * rows_per_wave;
  int64_t r_end = r_begin + rows_per_wave;
  if (r_end > P) r_end = P;
  const int n_slot = r_end > r_begin ? (int)((r_end - r_begin + 3) / 4) : 0;
  dk_d4 acc[DKD_C][4];
#pragma unroll
  for (int c = 0; c < DKD_C; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[c][e] = dk_d4{0., 0., 0., 0.};
  dkd_sweep<C, DKT_D>(XT, ldn, unit0 * 64, r_begin, n_slot, v, lane, acc);
  double g[4][4];
  dkd_fold(acc, img, wave, lane, g);
  const int i = lane & 15, k = lane >> 4;
  if (wave < C && i < K) {
    const double* rs = rowscale.p[i];
    double* o = out.p[i];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t row = (unit0 + wave) * 64 + 4 * (k + 4 * reg) + e;
        if (row < n) {
          const double t = g[e][reg];
          double w = t;
          if (rs) w *= rs[row];
          o[row * out_stride] = w;
          twt = fma(w, t, twt);
        }
      }
  }
  __syncthreads();   // the image is reused by the next sweep
}

__global__ __launch_bounds__(DKD_WAVES * WAVE) void dense_dot_kd_kernel(
    int K, int64_t n, int64_t P, int64_t ldn, const float* __restrict__ XT,
    const double* __restrict__ v, ChainPtrs rowscale, ChainOut out,
    int out_stride, double* __restrict__ twt_part,
    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char dk_smem[];
  __shared__ double s_twt[DKD_WAVES][16];
  double* img = reinterpret_cast<double*>(dk_smem);
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int64_t n_unit = ldn / 64;
  const int64_t base = n_unit / gridDim.x, extra = n_unit % gridDim.x;
  const int64_t b = blockIdx.x;
  int64_t u = b * base + (b < extra ? b : extra);
  const int64_t u1 = u + base + (b < extra ? 1 : 0);
  double twt = 0.;  // this lane's part of <t_c, Omega_c t_c>, c = lane & 15
  for (; u + 4 <= u1; u += 4)
    dkd_dot_units<4>(K, n, P, ldn, XT, v, rowscale, out, out_stride, u, img, wave, lane, twt);
  if (u1 - u == 3)
    dkd_dot_units<3>(K, n, P, ldn, XT, v, rowscale, out, out_stride, u, img, wave, lane, twt);
  else if (u1 - u == 2)
    dkd_dot_units<2>(K, n, P, ldn, XT, v, rowscale, out, out_stride, u, img, wave, lane, twt);
  else if (u1 - u == 1)
    dkd_dot_units<1>(K, n, P, ldn, XT, v, rowscale, out, out_stride, u, img, wave, lane, twt);
  if (twt_part) {
    // lanes i, i + 16, i + 32, i + 48 hold chain i's parts: fixed order
    double a = twt + __shfl_xor(twt, 16);
    a = a + __shfl_xor(a, 32);
    if (lane < 16) s_twt[wave][lane] = a;
    __syncthreads();
    if (tid < K) {
      double tot = 0.;
      for (int wv = 0; wv < DKD_WAVES; ++wv) tot += s_twt[wv][tid];
      twt_part[tid * NPART + blockIdx.x] = tot;
    }
  }
}

// XT[c][r] = X[r][c] for c < ld (rows of XT past P: X's zero padding columns;
// past ld and columns past n: zero)
__global__ __launch_bounds__(256) void dense_transpose_kernel(
    int64_t n, int64_t ld, int64_t ldn, int64_t xt_rows,
    const float* __restrict__ X, float* __restrict__ XT) {
  __shared__ float tile[64][65];
  const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int64_t r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < n && c < ld) ? X[r * ld + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int64_t c = c0 + j, r = r0 + tx;
    if (c < xt_rows && r < ldn) XT[c * ldn + r] = tile[tx][j];
  }
}

static int ensure_dense_transpose(bbx_design* h) {
  if (h->dense_xt.ptr) return BBX_OK;
  const int64_t ldn = (h->n + 63) / 64 * 64;
  const int64_t rows = (h->dense_ld + 63) / 64 * 64 + DENSE_PAD_ROWS;
  BBX_TRY(h->dense_xt.alloc(sizeof(float) * (size_t)rows * (size_t)ldn));
  BBX_HIP(hipMemsetAsync(h->dense_xt.ptr, 0,
                         sizeof(float) * (size_t)rows * (size_t)ldn, h->stream));
  const dim3 grid((unsigned)(ldn / 64), (unsigned)((h->dense_ld + 63) / 64));
  hipLaunchKernelGGL(dense_transpose_kernel, grid, dim3(256), 0, h->stream,
                     h->n, h->dense_ld, ldn, rows, h->dense.as<float>(),
                     h->dense_xt.as<float>());
  BBX_HIP(hipGetLastError());
  h->dense_xt_ld = ldn;
  return BBX_OK;
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
  BBX_HIP(hipFuncSetAttribute(
      reinterpret_cast<const void*>(&dense_tdot_kd_kernel),
      hipFuncAttributeMaxDynamicSharedMemorySize, DKD_IMG));
  BBX_HIP(hipFuncSetAttribute(
      reinterpret_cast<const void*>(&dense_dot_kd_kernel),
      hipFuncAttributeMaxDynamicSharedMemorySize, DKD_IMG));
  return BBX_OK;
}

int launch_dot_dense_k(bbx_design* h, int K, const double* d_v,
                       const TiledBatchArgs& ba, double* d_twt_part) {
  if (!dense_batch_applies(h))
    return fail(BBX_ERR_STATE, "batched dense products need f32 storage");
  BBX_TRY(dk_set_attr());
  if (DK_DIRECT) BBX_TRY(ensure_dense_transpose(h));
  h->n_dot += 1;
  BBX_TRY(timer_begin(h, 0));
  if (DK_DIRECT) {
    hipLaunchKernelGGL(dense_dot_kd_kernel, dim3(DK_DOT_WGS),
                       dim3(DKD_WAVES * WAVE), DKD_IMG, h->stream, K, h->n,
                       h->P, h->dense_xt_ld, h->dense_xt.as<float>(), d_v,
                       ba.rowscale, ba.out, ba.out_stride, d_twt_part,
                       h->skip_flag);
  } else {
    hipLaunchKernelGGL(dense_dot_k_kernel, dim3(DK_DOT_WGS),
                       dim3(DkDot::waves * WAVE), DkDot::lds_bytes, h->stream, K,
                       h->n, h->P, h->dense_ld, h->dense.as<float>(), d_v,
                       ba.rowscale, ba.out, ba.out_stride, d_twt_part,
                       h->skip_flag);
  }
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
  h->n_tdot += 1;
  BBX_TRY(timer_begin(h, 1));
  if (DK_DIRECT) {
    const int n_colblk = (int)((h->dense_ld + 64 * DKD_C - 1) / (64 * DKD_C));
    const int64_t parts = (int64_t)DK_TDOT_CHUNKS * DKD_WAVES;
    const int64_t rows_per_wave = ((h->n + parts - 1) / parts + 3) / 4 * 4;
    hipLaunchKernelGGL(dense_tdot_kd_kernel,
                       dim3((unsigned)(n_colblk * DK_TDOT_CHUNKS)),
                       dim3(DKD_WAVES * WAVE), DKD_IMG, h->stream, K, h->n,
                       h->dense_ld, rows_per_wave, h->dense.as<float>(), d_w,
                       h->dense_batch_slab.as<double>(), h->skip_flag);
  } else {
    const int n_colblk = (int)((h->dense_ld + DK_COLS - 1) / DK_COLS);
    const int64_t rows_per_chunk =
        ((h->n + DK_TDOT_CHUNKS - 1) / DK_TDOT_CHUNKS + DkTdot::rows - 1) /
        DkTdot::rows * DkTdot::rows;
    const int n_wave = n_colblk * DK_TDOT_CHUNKS;
    const unsigned grid = (unsigned)((n_wave + DkTdot::waves - 1) / DkTdot::waves);
    hipLaunchKernelGGL(dense_tdot_k_kernel, dim3(grid),
                       dim3(DkTdot::waves * WAVE), DkTdot::lds_bytes, h->stream,
                       K, h->n, h->dense_ld, rows_per_chunk, n_colblk,
                       h->dense.as<float>(), d_w,
                       h->dense_batch_slab.as<double>(), h->skip_flag);
  }
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
