// Dense design, batched chains: the two products of the CG operator for K <= 32
// right-hand sides at once, on the matrix cores.
//
// A GEMV cannot feed v_mfma_f64_16x16x4_f64 (one right-hand side: 15 of the 16
// columns of B are padding; measured 1.7x SLOWER than the vector ALUs,
// LABNOTES.md 3.3).  K chains that share the pass over X are the shape that can:
//   T[n x K] = X[n x P] V[P x K]          (dense_matrix.py:42, K at a time)
//   G[P x K] = X^T[P x n] W[n x K]        (dense_matrix.py:52, K at a time)
// with the chains in the 16 columns of B / D (32 chains: two B operands per A
// operand).  The single-chain path fuses the
// two products into one pass (dense.hip) because its slice of v and of the
// result fits a thread's registers; for K chains that state is 2 P K doubles
// per workgroup (512 KB at K = 4: a CU's whole register file), so the batch
// reads the matrix twice per operator application -- for all its chains.
//
// Operand maps (cdna_hip_programming.md "f64 MFMA"): A[l & 15][l >> 4],
// B[l >> 4][l & 15], D col = l & 15, row = (l >> 4) + 4 reg.
// Every column of D is computed from its own column of B only: a chain's
// numbers do not depend on the other chains of the batch (bit for bit).
#include "common.hpp"

// -DDK_ABLATE=2: plain arithmetic instead of the MFMAs; =3: the loads stay on the first rows
// (cache hits: matrix cores + issue without the HBM stream); =4: no loads at all after the
// ring's priming -- timing only, wrong results
#ifndef DK_ABLATE
#define DK_ABLATE 0
#endif

namespace bbx {

typedef double dk_d4 __attribute__((ext_vector_type(4)));
constexpr int DK_KS = DENSE_BATCH_STRIDE;   // 16: the columns of one B operand; a batch of
                                            // 32 chains interleaves two of them (stride 32)
constexpr int DK_DOT_WGS = 256;             // one persistent workgroup per CU
constexpr int DK_TDOT_CHUNKS = 8;           // row chunks of the transposed product

// ---------------------------------------------------------------------------
// Direct forms: the matrix goes from HBM straight into the A-operand registers.
//
// The first version of these kernels staged 64 x 64 tiles through LDS by
// LDS-DMA (scripts/experiments/r03_dense_lds_stages.patch): a wave's share of
// LDS held two stages, so ONE was in flight while the other fed the matrix
// cores, and every stage paid an LDS round trip before its first MFMA: 1.57 /
// 1.37 ms per product where the HBM pass is 1.0-1.1 ms on these boxes and the
// MFMAs 0.66 ms.  A wave's registers (512 per lane at one wave per SIMD) hold
// three times what its LDS share does (an LDS ring in the same place was also
// tried: scripts/experiments/r03_dense_lds_ring.hip.txt), and one orientation
// needs no staging at all:
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
// waits (the idiom of spmv_tiled.hip).  Reads may run past a wave's rows or the
// end of the matrix (the allocations are padded with zero rows); what they fetch
// meets a zero B operand or a column that is never stored.
//
// What bounds the passes is the ISSUE of the matrix cores (-DDK_ABLATE=3 / 4:
// without the HBM stream they take 85 % of their time), so:
//  * accumulators live in ARCHITECTURAL VGPRs: with a[..] accumulators
//    v_mfma_f64_16x16x4_f64 issues at 58-64 ns per MFMA per SIMD, with v[..] at
//    27 ns (scripts/probes/mfma_f64_acc.hip).  128 registers of tiles, 18 per
//    ring slot, and the rest have to fit the 256 VGPRs -- hence dkd_cw();
//  * a statement is a UNIT (four MFMAs) with the next unit's conversions in
//    front of it (dk_unit_f32): see there.
#ifndef DKT_D
#define DKT_D 5      // ring depth (slots of 4 rows): 18 VGPRs per slot next to 128 of accumulators
#endif
#ifndef DK_UNIT_ASM
#define DK_UNIT_ASM 1   // f32: four MFMAs per asm statement, conversions one unit ahead
#endif
#ifndef DKT_D2
#define DKT_D2 6     // ... with two groups of chains (two units per sweep at f32: 12 VGPRs per slot)
#endif
#ifndef DKT_D2_F64
#define DKT_D2_F64 5 // ... and f64 storage: four units per sweep, 20 VGPRs per slot -- at depth 6
                     // the X V kernel sat at 256 VGPRs and parked 16 loop invariants in AGPRs;
                     // depths 5-12 run equally fast (tests/test_kernel_resources.py)
#endif

typedef float dk_f4 __attribute__((ext_vector_type(4)));

template <int IMM, typename V>
__device__ __forceinline__ void dk_ld_x4(V& dst, unsigned voff,
                                         const void* sbase) {
  static_assert(sizeof(V) == 16, "one 16-byte load per lane");
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt"
               : "=v"(dst)
               : "v"(voff), "s"(sbase), "n"(IMM)
               : "memory");
}
template <int IMM>
__device__ __forceinline__ void dk_ld_d(double& dst, unsigned voff,
                                        const void* sbase) {
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3"
               : "=v"(dst)
               : "v"(voff), "s"(sbase), "n"(IMM)
               : "memory");
}

// One MFMA per statement (f32 storage): conversion of the A operand, the wait
// states between a VALU result and the matrix core's read of it, the MFMA.
// Kept for the ablation builds and as the reference form; the kernels use
// dk_unit_f32 below (-DDK_UNIT_ASM=0 selects this one: 45 ns per MFMA).
//  * The MFMAs are written in asm with the accumulator as an explicit operand
//    ("+v": architectural VGPRs, see the file header) because through the
//    builtin the compiler picks AGPR accumulators, carries them across the loop
//    boundary in VGPRs and copies all of them in and out every period.
//  * An MFMA written in asm is invisible to the compiler's hazard recogniser:
//    with the conversion left outside, `v_cvt_f64_f32 v[a:b], ..` directly in
//    front of `v_mfma .., v[a:b], ..` read the register's OLD contents (there
//    is no interlock; every column but one of the first version was wrong).
//    The s_nop below is that spacing; it also covers a B operand produced by
//    the VALU just before the statement.  The rest holds by construction: an
//    accumulator is touched again only after >= 3 other MFMAs (the statements
//    are volatile and keep their source order), and the epilogue's reads are
//    preceded by explicit s_nops.
__device__ __forceinline__ void dk_mfma(dk_d4& acc, float x, double b) {
  double t;
  asm volatile(
      "v_cvt_f64_f32 %1, %2\n\ts_nop 3\n\t"
      "v_mfma_f64_16x16x4_f64 %0, %1, %3, %0"
      : "+v"(acc), "=&v"(t)
      : "v"(x), "v"(b));
}

// Two groups of 16 chains: one conversion of the A operand feeds both MFMAs
__device__ __forceinline__ void dk_mfma2(dk_d4& acc0, dk_d4& acc1, float x,
                                         double b0, double b1) {
  double t;
  asm volatile(
      "v_cvt_f64_f32 %2, %3\n\ts_nop 3\n\t"
      "v_mfma_f64_16x16x4_f64 %0, %2, %4, %0\n\t"
      "v_mfma_f64_16x16x4_f64 %1, %2, %5, %1"
      : "+v"(acc0), "+v"(acc1), "=&v"(t)
      : "v"(x), "v"(b0), "v"(b1));
}

// f64 storage: the lane's 16 bytes are two doubles, fed to the MFMAs as they
// are (no VALU between the counted load and the matrix core; the s_nop spaces
// a B operand the VALU may have just produced).
__device__ __forceinline__ void dk_mfma(dk_d4& acc, double x, double b) {
  asm volatile("s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0"
               : "+v"(acc)
               : "v"(x), "v"(b));
}
__device__ __forceinline__ void dk_mfma2(dk_d4& acc0, dk_d4& acc1, double x,
                                         double b0, double b1) {
  asm volatile(
      "s_nop 3\n\tv_mfma_f64_16x16x4_f64 %0, %2, %3, %0\n\t"
      "v_mfma_f64_16x16x4_f64 %1, %2, %4, %1"
      : "+v"(acc0), "+v"(acc1)
      : "v"(x), "v"(b0), "v"(b1));
}

// f32 storage, one UNIT (the four MFMAs of a lane's float4) per asm statement,
// software-pipelined: the conversions of the NEXT unit sit in front of this
// unit's MFMAs.  One statement per MFMA (conversion, wait states, MFMA) ran the
// matrix cores at 45 ns per MFMA instead of the 27 ns they can sustain -- with
// or without the HBM stream (-DDK_ABLATE=3): every MFMA waited for its own
// conversion, and the conversion reused the register the previous MFMA was
// still reading.  Here the operands of a unit were converted a whole unit (four
// MFMAs) earlier into registers of their own.
#define DK_MF(A, B, E) \
  "v_mfma_f64_16x16x4_f64 %[" A #E "], %[c" #E "], %[" B "], %[" A #E "]\n\t"
#define DK_CV(E) "v_cvt_f64_f32 %[n" #E "], %[x" #E "]\n\t"
#ifndef DK_UNIT_ORDER
#define DK_UNIT_ORDER 0   // 0: the four conversions, then the MFMAs; 1: interleaved
#endif
#if DK_UNIT_ORDER == 0
#define DK_UNIT1_NEXT DK_CV(0) DK_CV(1) DK_CV(2) DK_CV(3) DK_MF("p", "b0", 0) \
                      DK_MF("p", "b0", 1) DK_MF("p", "b0", 2) DK_MF("p", "b0", 3)
#define DK_UNIT2_NEXT                                                         \
  DK_CV(0) DK_CV(1) DK_CV(2) DK_CV(3) DK_MF("p", "b0", 0) DK_MF("q", "b1", 0) \
  DK_MF("p", "b0", 1) DK_MF("q", "b1", 1) DK_MF("p", "b0", 2)                 \
  DK_MF("q", "b1", 2) DK_MF("p", "b0", 3) DK_MF("q", "b1", 3)
#else
#define DK_UNIT1_NEXT "s_nop 1\n\t" DK_MF("p", "b0", 0) DK_CV(0) DK_MF("p", "b0", 1) DK_CV(1) \
                      DK_MF("p", "b0", 2) DK_CV(2) DK_MF("p", "b0", 3) DK_CV(3)
#define DK_UNIT2_NEXT                                                         \
  "s_nop 1\n\t" DK_MF("p", "b0", 0) DK_MF("q", "b1", 0) DK_CV(0) DK_MF("p", "b0", 1)        \
  DK_MF("q", "b1", 1) DK_CV(1) DK_MF("p", "b0", 2) DK_MF("q", "b1", 2)        \
  DK_CV(2) DK_MF("p", "b0", 3) DK_MF("q", "b1", 3) DK_CV(3)
#endif
#define DK_UNIT1_LAST DK_MF("p", "b0", 0) DK_MF("p", "b0", 1) DK_MF("p", "b0", 2) \
                      DK_MF("p", "b0", 3)
#define DK_UNIT2_LAST                                                         \
  DK_MF("p", "b0", 0) DK_MF("q", "b1", 0) DK_MF("p", "b0", 1)                 \
  DK_MF("q", "b1", 1) DK_MF("p", "b0", 2) DK_MF("q", "b1", 2)                 \
  DK_MF("p", "b0", 3) DK_MF("q", "b1", 3)
template <int NG, bool NEXT>
__device__ __forceinline__ void dk_unit_f32(dk_d4 (&p)[4], dk_d4 (&q)[4],
                                            const double (&tc)[4],
                                            double (&tn)[4], const dk_f4& xn,
                                            double b0, double b1) {
  if (NG == 1 && NEXT) {
    asm volatile(DK_UNIT1_NEXT
                 : [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]),
                   [n0] "=&v"(tn[0]), [n1] "=&v"(tn[1]), [n2] "=&v"(tn[2]), [n3] "=&v"(tn[3])
                 : [c0] "v"(tc[0]), [c1] "v"(tc[1]), [c2] "v"(tc[2]), [c3] "v"(tc[3]),
                   [x0] "v"(xn[0]), [x1] "v"(xn[1]), [x2] "v"(xn[2]), [x3] "v"(xn[3]),
                   [b0] "v"(b0));
  } else if (NG == 1) {
    asm volatile("s_nop 3\n\t" DK_UNIT1_LAST
                 : [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3])
                 : [c0] "v"(tc[0]), [c1] "v"(tc[1]), [c2] "v"(tc[2]), [c3] "v"(tc[3]),
                   [b0] "v"(b0));
  } else if (NEXT) {
    asm volatile(DK_UNIT2_NEXT
                 : [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]),
                   [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]),
                   [n0] "=&v"(tn[0]), [n1] "=&v"(tn[1]), [n2] "=&v"(tn[2]), [n3] "=&v"(tn[3])
                 : [c0] "v"(tc[0]), [c1] "v"(tc[1]), [c2] "v"(tc[2]), [c3] "v"(tc[3]),
                   [x0] "v"(xn[0]), [x1] "v"(xn[1]), [x2] "v"(xn[2]), [x3] "v"(xn[3]),
                   [b0] "v"(b0), [b1] "v"(b1));
  } else {
    asm volatile("s_nop 3\n\t" DK_UNIT2_LAST
                 : [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]),
                   [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3])
                 : [c0] "v"(tc[0]), [c1] "v"(tc[1]), [c2] "v"(tc[2]), [c3] "v"(tc[3]),
                   [b0] "v"(b0), [b1] "v"(b1));
  }
}
#undef DK_MF
#undef DK_CV
#undef DK_UNIT1_NEXT
#undef DK_UNIT1_LAST
#undef DK_UNIT2_NEXT
#undef DK_UNIT2_LAST

constexpr int DKD_WAVES = 4;        // one per SIMD: 512 registers per lane
constexpr int DKD_C = 4;            // most units (one load instruction wide) a sweep carries
// Units per sweep: the accumulators (NG groups x CW units x E tiles x 8 registers)
// have to fit 128 ARCHITECTURAL VGPRs -- with its accumulators in AGPRs
// v_mfma_f64_16x16x4_f64 issues at 58-64 ns per MFMA per SIMD, in VGPRs at 27 ns
// (scripts/probes/mfma_f64_acc.hip, profiles/r03_mfma_f64_acc.txt)
constexpr int dkd_cw(int NG, int E) { return NG * E <= 4 ? 4 : 2; }
// Per storage type T: a lane's 16 bytes are E elements, a load instruction
// covers 4 rows x U = 16 E columns (a "unit": 64 columns of f32, 32 of f64)
template <typename T>
struct DkT {
  static constexpr int E = 16 / (int)sizeof(T);
  static constexpr int U = 16 * E;
  typedef T vec __attribute__((ext_vector_type(16 / sizeof(T))));
};
// ring depth of one instantiation
template <typename T, int NG>
constexpr int dkd_depth() {
  return NG == 1 ? DKT_D : (sizeof(T) == 8 ? DKT_D2_F64 : DKT_D2);
}
constexpr int DKD_IMG = DKD_WAVES * DKD_C * 16 * WAVE * 8;  // 128 KB of LDS at most
// the fold image of one instantiation: [waves][CW][E][4][64] doubles
template <typename T, int NG>
constexpr int dkd_img() {
  return DKD_WAVES * dkd_cw(NG, DkT<T>::E) * DkT<T>::E * 4 * WAVE * 8;
}

// One wave's sweep: acc[g][c][e] += sum over rows [r_begin, r_begin + 4 n_slot)
// of M[r][col0 + U c + E i' + e] * B[r][16 g + chain], i' the MFMA's row index.
// (A wave without rows primes its ring on the first rows and computes
// nothing; the ring's read-ahead relies on the padding of M and B.)
template <typename T, int C, int D, int NG>
__device__ __forceinline__ void dkd_sweep(
    const T* __restrict__ M, int64_t ldm, int64_t col0, int64_t r_begin,
    int n_slot, const double* __restrict__ Bop, int lane,
    dk_d4 (&acc)[NG][dkd_cw(NG, DkT<T>::E)][DkT<T>::E]) {
  constexpr int CW = dkd_cw(NG, DkT<T>::E);
  static_assert(C >= 1 && C <= CW, "units per sweep");
  static_assert(NG == 1 || NG == 2, "groups of 16 chains");
  constexpr int E = DkT<T>::E;
  typedef typename DkT<T>::vec vec_t;
  constexpr int KS = NG * DK_KS;
  constexpr int LPS = C + NG;
  constexpr int WAITC = (D - 1) * LPS;
  static_assert(WAITC < 64, "vmcnt is a 6-bit field");
  const int i = lane & 15, k = lane >> 4;
  const int n_period = (n_slot + D - 1) / D;
  const unsigned voff_a = (unsigned)(((int64_t)k * ldm + E * i) * sizeof(T));
  const unsigned voff_b = (unsigned)((k * KS + i) * 8);
  const int64_t r_ring = n_slot > 0 ? r_begin : 0;
  const T* sa = M + r_ring * ldm + col0;         // wave-uniform, 4 rows per slot
  const double* sb = Bop + r_ring * KS;
  vec_t xa[D][C];
  double bw[D][NG];
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing of the compiler's in the queue
#define DKD_ISSUE(KK)                                                         \
  do {                                                                        \
    dk_ld_x4<0>(xa[KK][0], voff_a, sa);                                       \
    if (C > 1) dk_ld_x4<256>(xa[KK][C > 1 ? 1 : 0], voff_a, sa);              \
    if (C > 2) dk_ld_x4<512>(xa[KK][C > 2 ? 2 : 0], voff_a, sa);              \
    if (C > 3) dk_ld_x4<768>(xa[KK][C > 3 ? 3 : 0], voff_a, sa);              \
    dk_ld_d<0>(bw[KK][0], voff_b, sb);                                        \
    if (NG > 1) dk_ld_d<DK_KS * 8>(bw[KK][NG - 1], voff_b, sb);               \
    if (DK_ABLATE != 3) { /* 3: every slot re-reads the first rows (cache) */ \
      sa += 4 * ldm;                                                          \
      sb += 4 * KS;                                                           \
    }                                                                         \
  } while (0)
#define DKD_TIE(KK)                                                           \
  do {                                                                        \
    _Pragma("unroll") for (int c = 0; c < C; ++c)                             \
        asm volatile("" : "+v"(xa[KK][c]));                                   \
    _Pragma("unroll") for (int g = 0; g < NG; ++g)                            \
        asm volatile("" : "+v"(bw[KK][g]));                                   \
  } while (0)
#pragma unroll
  for (int kk = 0; kk < D; ++kk) DKD_ISSUE(kk);
  if constexpr (E == 4 && DK_ABLATE != 2 && DK_UNIT_ASM) {
    // f32: a unit (four MFMAs) per asm statement with the conversions of the
    // NEXT unit between its MFMAs -- across slots too, so the next slot is
    // waited for one step early (D - 2 slots stay in flight)
    static_assert(D >= 3, "the unit pipeline looks one slot ahead");
    constexpr int WAITN = (D - 2) * LPS;
    double tc[4], tn[4];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITC) : "memory");
    DKD_TIE(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) tc[e] = (double)xa[0][0][e];
    asm volatile("s_nop 3" ::: "memory");   // VALU result -> first MFMA
    for (int p = 0; p < n_period; ++p) {
#pragma unroll
      for (int kk = 0; kk < D; ++kk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        DKD_TIE((kk + 1) % D);
        const bool live = p * D + kk < n_slot;
        const double b0 = live ? bw[kk][0] : 0.;
        const double b1 = live ? bw[kk][NG - 1] : 0.;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const vec_t& xn = (c + 1 < C) ? xa[kk][c + 1 < C ? c + 1 : 0]
                                        : xa[(kk + 1) % D][0];
          dk_unit_f32<NG, true>(acc[0][c], acc[NG - 1][c], tc, tn, xn, b0, b1);
#pragma unroll
          for (int e = 0; e < 4; ++e) tc[e] = tn[e];
        }
        if (DK_ABLATE != 4) DKD_ISSUE(kk);   // 4: no loads after the priming
      }
    }
  } else {
  for (int p = 0; p < n_period; ++p) {
#pragma unroll
    for (int kk = 0; kk < D; ++kk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITC) : "memory");
      DKD_TIE(kk);
      // a slot past this wave's rows belongs to the next wave (or is
      // padding): it meets a zero B operand, no branch between the MFMAs
      const bool live = p * D + kk < n_slot;
      const double b0 = live ? bw[kk][0] : 0.;
      const double b1 = live ? bw[kk][NG - 1] : 0.;
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if (DK_ABLATE == 2) acc[0][c][e][0] += (double)xa[kk][c][e] * (b0 + b1);
          else if (NG == 1) dk_mfma(acc[0][c][e], xa[kk][c][e], b0);
          else dk_mfma2(acc[0][c][e], acc[NG - 1][c][e], xa[kk][c][e], b0, b1);
        }
      DKD_ISSUE(kk);
    }
  }
  }
  // the D slots issued past the end: landed before their registers are free
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int kk = 0; kk < D; ++kk) DKD_TIE(kk);
#undef DKD_ISSUE
#undef DKD_TIE
  // 18 wait states between the last MFMA and a read of its result
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0][0]));
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (g + c + e > 0) asm volatile("" : "+v"(acc[g][c][e]));
}

// The four waves' accumulators through LDS, added in the fixed order
// (w0 + w1) + (w2 + w3); wave c' returns unit c' in g[e][reg]:
// g[e][reg] = G[col0 + U c' + E ((lane >> 4) + 4 reg) + e][chain lane & 15].
template <int E, int CW>
__device__ __forceinline__ void dkd_fold(const dk_d4 (&acc)[CW][E],
                                         double* img, int wave, int lane,
                                         double (&g)[E][4]) {
#pragma unroll
  for (int c = 0; c < CW; ++c)
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        img[((((wave * CW + c) * E + e) * 4 + reg) << 6) + lane] = acc[c][e][reg];
  __syncthreads();
  constexpr int WS = CW * E * 4 * WAVE;
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      // (waves past the sweep's units read a neighbour's slot: never stored)
      const int o = ((((wave % CW) * E + e) * 4 + reg) << 6) + lane;
      g[e][reg] = (img[o] + img[WS + o]) + (img[2 * WS + o] + img[3 * WS + o]);
    }
  __syncthreads();   // the image is free again (next group of chains, next sweep)
}

// X^T W: a workgroup = (block of C units of columns, one of DK_TDOT_CHUNKS row
// chunks); its four waves take quarters of the chunk's rows.  blockIdx.x % 8 =
// the chunk: under round-robin placement the workgroups of an XCD share a row
// chunk, and its slice of W streams through that XCD's L2 once.
template <typename T, int NG>
__global__ __launch_bounds__(DKD_WAVES * WAVE) void dense_tdot_kd_kernel(
    int K, int64_t n, int64_t ld, int64_t rows_per_wave,
    const T* __restrict__ X, const double* __restrict__ w,
    double* __restrict__ slab, const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  constexpr int E = DkT<T>::E, U = DkT<T>::U, CW = dkd_cw(NG, E);
  extern __shared__ __attribute__((aligned(16))) unsigned char dk_smem[];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int chunk = (int)(blockIdx.x % DK_TDOT_CHUNKS);
  const int colblk = (int)(blockIdx.x / DK_TDOT_CHUNKS);
  const int64_t col0 = (int64_t)colblk * (U * CW);
  const int64_t r_begin =
      ((int64_t)chunk * DKD_WAVES + wave) * rows_per_wave;  // multiple of 4
  int64_t r_end = r_begin + rows_per_wave;
  if (r_end > n) r_end = n;
  const int n_slot = r_end > r_begin ? (int)((r_end - r_begin + 3) / 4) : 0;
  dk_d4 acc[NG][CW][E];
#pragma unroll
  for (int gq = 0; gq < NG; ++gq)
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
      for (int e = 0; e < E; ++e) acc[gq][c][e] = dk_d4{0., 0., 0., 0.};
  dkd_sweep<T, CW, dkd_depth<T, NG>(), NG>(X, ld, col0, r_begin, n_slot, w,
                                           lane, acc);
  const int i = lane & 15, k = lane >> 4;
#pragma unroll
  for (int gq = 0; gq < NG; ++gq) {
    double g[E][4];
    dkd_fold<E, CW>(acc[gq], reinterpret_cast<double*>(dk_smem), wave, lane, g);
    const int chain = 16 * gq + i;
    if (wave < CW && chain < K) {
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int64_t col = col0 + U * wave + E * (k + 4 * reg) + e;
          if (col < ld)
            slab[((int64_t)chunk * ld + col) * (NG * DK_KS) + chain] = g[e][reg];
        }
    }
  }
}

// X V through the transposed copy: T = (X^T)^T V, XT row-major [P + pad][ldn].
// 256 persistent workgroups; workgroup b owns the units [u0, u1) of result
// rows (12 or 13 of them at 200 000 rows of f32) and sweeps all P rows of XT
// once per four units, its waves taking quarters of P; the tail of fewer than
// four units is a narrower sweep.  Epilogue per unit: rowscale, the store,
// <t, Omega t>.
template <typename T, int C, int NG>
__device__ __forceinline__ void dkd_dot_units(
    int K, int64_t n, int64_t P, int64_t ldn, const T* __restrict__ XT,
    const double* __restrict__ v, const ChainPtrs& rowscale, const ChainOut& out,
    int out_stride, int64_t unit0, double* img, int wave, int lane,
    double (&twt)[NG]) {
  constexpr int E = DkT<T>::E, U = DkT<T>::U, CW = dkd_cw(NG, E);
  const int64_t rows_per_wave = ((P + DKD_WAVES - 1) / DKD_WAVES + 3) / 4 * 4;
  const int64_t r_begin = (int64_t)wave * rows_per_wave;
  int64_t r_end = r_begin + rows_per_wave;
  if (r_end > P) r_end = P;
  const int n_slot = r_end > r_begin ? (int)((r_end - r_begin + 3) / 4) : 0;
  dk_d4 acc[NG][CW][E];
#pragma unroll
  for (int gq = 0; gq < NG; ++gq)
#pragma unroll
    for (int c = 0; c < CW; ++c)
#pragma unroll
      for (int e = 0; e < E; ++e) acc[gq][c][e] = dk_d4{0., 0., 0., 0.};
  dkd_sweep<T, C, dkd_depth<T, NG>(), NG>(XT, ldn, unit0 * U, r_begin, n_slot,
                                          v, lane, acc);
  const int i = lane & 15, k = lane >> 4;
#pragma unroll
  for (int gq = 0; gq < NG; ++gq) {
    double g[E][4];
    dkd_fold<E, CW>(acc[gq], img, wave, lane, g);
    const int chain = 16 * gq + i;
    if (wave < C && chain < K) {
      const double* rs = rowscale.p[chain];
      double* o = out.p[chain];
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int64_t row = (unit0 + wave) * U + E * (k + 4 * reg) + e;
          if (row < n) {
            const double t = g[e][reg];
            double w = t;
            if (rs) w *= rs[row];
            o[row * out_stride] = w;
            twt[gq] = fma(w, t, twt[gq]);
          }
        }
    }
  }
}

template <typename T, int NG>
__global__ __launch_bounds__(DKD_WAVES * WAVE) void dense_dot_kd_kernel(
    int K, int64_t n, int64_t P, int64_t ldn, const T* __restrict__ XT,
    const double* __restrict__ v, ChainPtrs rowscale, ChainOut out,
    int out_stride, double* __restrict__ twt_part,
    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char dk_smem[];
  __shared__ double s_twt[DKD_WAVES][NG * 16];
  double* img = reinterpret_cast<double*>(dk_smem);
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int64_t n_unit = ldn / DkT<T>::U;
  const int64_t base = n_unit / gridDim.x, extra = n_unit % gridDim.x;
  const int64_t b = blockIdx.x;
  int64_t u = b * base + (b < extra ? b : extra);
  const int64_t u1 = u + base + (b < extra ? 1 : 0);
  // this lane's parts of <t_c, Omega_c t_c>, c = 16 g + (lane & 15)
  double twt[NG];
#pragma unroll
  for (int gq = 0; gq < NG; ++gq) twt[gq] = 0.;
#define DKD_UNITS(C)                                                          \
  dkd_dot_units<T, C, NG>(K, n, P, ldn, XT, v, rowscale, out, out_stride, u,  \
                          img, wave, lane, twt)
  constexpr int CW = dkd_cw(NG, DkT<T>::E);
  for (; u + CW <= u1; u += CW) DKD_UNITS(CW);
  if constexpr (CW == 4) {
    if (u1 - u == 3) DKD_UNITS(3);
    else if (u1 - u == 2) DKD_UNITS(2);
    else if (u1 - u == 1) DKD_UNITS(1);
  } else {
    if (u1 - u == 1) DKD_UNITS(1);
  }
#undef DKD_UNITS
  if (twt_part) {
    // lanes i, i + 16, i + 32, i + 48 hold chain i's parts: fixed order
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      double a = twt[gq] + __shfl_xor(twt[gq], 16);
      a = a + __shfl_xor(a, 32);
      if (lane < 16) s_twt[wave][16 * gq + lane] = a;
    }
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
template <typename T>
__global__ __launch_bounds__(256) void dense_transpose_kernel(
    int64_t n, int64_t ld, int64_t ldn, int64_t xt_rows,
    const T* __restrict__ X, T* __restrict__ XT) {
  __shared__ T tile[64][65];
  const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int j = ty; j < 64; j += 4) {
    const int64_t r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < n && c < ld) ? X[r * ld + c] : (T)0;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int64_t c = c0 + j, r = r0 + tx;
    if (c < xt_rows && r < ldn) XT[c * ldn + r] = tile[tx][j];
  }
}

template <typename T>
static int ensure_dense_transpose_t(bbx_design* h) {
  const int64_t ldn = (h->n + 63) / 64 * 64;
  const int64_t rows = (h->dense_ld + 63) / 64 * 64 + DENSE_PAD_ROWS;
  BBX_TRY(h->dense_xt.alloc(sizeof(T) * (size_t)rows * (size_t)ldn));
  BBX_HIP(hipMemsetAsync(h->dense_xt.ptr, 0,
                         sizeof(T) * (size_t)rows * (size_t)ldn, h->stream));
  const dim3 grid((unsigned)(ldn / 64), (unsigned)((h->dense_ld + 63) / 64));
  BBX_LAUNCH(dense_transpose_kernel<T>, grid, dim3(256), 0, h->stream,
                     h->n, h->dense_ld, ldn, rows, h->dense.as<T>(),
                     h->dense_xt.as<T>());
  BBX_HIP(hipGetLastError());
  h->dense_xt_ld = ldn;
  return BBX_OK;
}
static int ensure_dense_transpose(bbx_design* h) {
  if (h->dense_xt.ptr) return BBX_OK;
  return h->dense_dtype == BBX_F32 ? ensure_dense_transpose_t<float>(h)
                                   : ensure_dense_transpose_t<double>(h);
}

bool dense_batch_applies(const bbx_design* h) {
  // rows start on 16-byte boundaries: ld is a multiple of 8 elements
  return !h->sparse && h->dense_ld % 8 == 0;
}

static int dk_set_attr(bbx_design* h) {
  if (h->dense_batch_attr) return BBX_OK;   // per design, i.e. per device
  for (const void* f :
       {reinterpret_cast<const void*>(&dense_tdot_kd_kernel<float, 1>),
        reinterpret_cast<const void*>(&dense_tdot_kd_kernel<float, 2>),
        reinterpret_cast<const void*>(&dense_dot_kd_kernel<float, 1>),
        reinterpret_cast<const void*>(&dense_dot_kd_kernel<float, 2>),
        reinterpret_cast<const void*>(&dense_tdot_kd_kernel<double, 1>),
        reinterpret_cast<const void*>(&dense_tdot_kd_kernel<double, 2>),
        reinterpret_cast<const void*>(&dense_dot_kd_kernel<double, 1>),
        reinterpret_cast<const void*>(&dense_dot_kd_kernel<double, 2>)})
    BBX_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                DKD_IMG));
  h->dense_batch_attr = true;
  return BBX_OK;
}

int dense_batch_stride(int K) { return K > DK_KS ? 2 * DK_KS : DK_KS; }

int launch_dot_dense_k(bbx_design* h, int K, const double* d_v,
                       const TiledBatchArgs& ba, double* d_twt_part) {
  if (!dense_batch_applies(h))
    return fail(BBX_ERR_STATE, "batched dense products: unsupported layout");
  BBX_TRY(dk_set_attr(h));
  BBX_TRY(ensure_dense_transpose(h));
  h->n_dot += 1;
  BBX_TRY(timer_begin(h, 0));
#define DK_LAUNCH_DOT(T, NG)                                                  \
  BBX_LAUNCH((dense_dot_kd_kernel<T, NG>), dim3(DK_DOT_WGS),          \
                     dim3(DKD_WAVES* WAVE), DKD_IMG, h->stream, K, h->n, h->P, \
                     h->dense_xt_ld, h->dense_xt.as<T>(), d_v, ba.rowscale,   \
                     ba.out, ba.out_stride, d_twt_part, h->skip_flag)
  const bool f32 = h->dense_dtype == BBX_F32;
  if (f32 && K > DK_KS) DK_LAUNCH_DOT(float, 2);
  else if (f32) DK_LAUNCH_DOT(float, 1);
  else if (K > DK_KS) DK_LAUNCH_DOT(double, 2);
  else DK_LAUNCH_DOT(double, 1);
#undef DK_LAUNCH_DOT
  BBX_HIP(hipGetLastError());
  return timer_end(h, 0);
}

int launch_tdot_dense_k(bbx_design* h, int K, const double* d_w,
                        const double** slab, int* G) {
  if (!dense_batch_applies(h))
    return fail(BBX_ERR_STATE, "batched dense products: unsupported layout");
  BBX_TRY(dk_set_attr(h));
  const size_t need = sizeof(double) * (size_t)DK_TDOT_CHUNKS *
                      (size_t)h->dense_ld * (size_t)dense_batch_stride(K);
  if (h->dense_batch_slab.bytes < need) {
    BBX_TRY(h->dense_batch_slab.alloc(need));
    // columns past the batch's chains are never written: keep them zero
    BBX_HIP(hipMemsetAsync(h->dense_batch_slab.ptr, 0, need, h->stream));
  }
  const bool f32 = h->dense_dtype == BBX_F32;
  const int unit = f32 ? DkT<float>::U : DkT<double>::U;
  const int cw = dkd_cw(K > DK_KS ? 2 : 1, f32 ? DkT<float>::E : DkT<double>::E);
  const int n_colblk = (int)((h->dense_ld + unit * cw - 1) / (unit * cw));
  const int64_t parts = (int64_t)DK_TDOT_CHUNKS * DKD_WAVES;
  const int64_t rows_per_wave = ((h->n + parts - 1) / parts + 3) / 4 * 4;
  h->n_tdot += 1;
  BBX_TRY(timer_begin(h, 1));
#define DK_LAUNCH_TDOT(T, NG)                                                 \
  BBX_LAUNCH((dense_tdot_kd_kernel<T, NG>),                           \
                     dim3((unsigned)(n_colblk * DK_TDOT_CHUNKS)),             \
                     dim3(DKD_WAVES* WAVE), (dkd_img<T, NG>()), h->stream, K, h->n, \
                     h->dense_ld, rows_per_wave, h->dense.as<T>(), d_w,       \
                     h->dense_batch_slab.as<double>(), h->skip_flag)
  if (f32 && K > DK_KS) DK_LAUNCH_TDOT(float, 2);
  else if (f32) DK_LAUNCH_TDOT(float, 1);
  else if (K > DK_KS) DK_LAUNCH_TDOT(double, 2);
  else DK_LAUNCH_TDOT(double, 1);
#undef DK_LAUNCH_TDOT
  BBX_HIP(hipGetLastError());
  BBX_TRY(timer_end(h, 1));
  *slab = h->dense_batch_slab.as<double>();
  *G = DK_TDOT_CHUNKS;
  return BBX_OK;
}

// The cost model's view of a dense batch (measured figures, LABNOTES.md 3.6): one
// chain applies the operator in ONE pass over the matrix where the single-pass
// kernel applies (0.93 ms at 200k x 8k f32, 2.31 ms in f64), a batch needs
// TWO passes for all its chains together -- stream-bound up to 16 chains
// (1.05 ms each in f32, 1.97 ms in f64), bound by the f64 matrix cores at 32
// (1.64 / 2.06 ms).  Two f32 chains therefore run faster one after the other.
int dense_batch_predict(const bbx_design* h, int K, double* speedup) {
  const bool f32 = h->dense_dtype == BBX_F32;
  const double single = dense_fused_applies(h) ? (f32 ? 0.93 : 2.31)
                                               : (f32 ? 2.0 : 4.4);
  const double pass = K <= DK_KS ? (f32 ? 1.05 : 1.97) : (f32 ? 1.64 : 2.06);
  *speedup = (double)K * single / (2. * pass);
  return BBX_OK;
}

int dense_batch_bytes(const bbx_design* h, int K, int64_t* dot_bytes,
                      int64_t* tdot_bytes) {
  // the matrix (X or its transposed copy) once + the 16-column operands
  // (padding columns are read too) + what the chains' columns write
  const int64_t mat = h->n * h->dense_ld * (h->dense_dtype == BBX_F32 ? 4 : 8);
  const int64_t ks = dense_batch_stride(K);
  *dot_bytes = mat + 8 * ks * h->dense_ld + 8 * (int64_t)K * h->n;
  *tdot_bytes = mat + 8 * ks * h->n +
                8 * (int64_t)K * DK_TDOT_CHUNKS * h->dense_ld;
  return BBX_OK;
}

}  // namespace bbx
