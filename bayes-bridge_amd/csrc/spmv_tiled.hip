// LDS-tiled sparse operator (BBX_FORMAT_TILED).
//
// Why: with the reference CSR layout every stored entry costs one random 8-byte
// gather from an L2/Infinity-Cache resident vector (400 KB for X v, 8 MB for
// X^T w at the headline size).  Each gather drags a 128-byte line through the
// CU's L1, so the kernels run at the cache-line rate of the memory pipeline,
// ~10x below the HBM roofline of the index stream (measured: profiles/).  Here
// the vector slice a workgroup needs sits in LDS (160 KB per CU) and the matrix
// is re-blocked so that every gather is an LDS read:
//
//   * columns are cut into blocks of W <= 16128 so that a slice of the input
//     vector (W doubles) fits in LDS next to PR row accumulators;
//   * rows are cut into panels of PR rows; tile = (panel, column block);
//   * inside a tile the non-empty rows are sorted by their entry count and
//     grouped 128 at a time into slices (one wavefront each, two rows per
//     lane), stored "sliced ELL": [step q][lane][row A: 4 ids | row B: 4 ids]
//     block-local uint16 column ids = one 16-byte load per lane and step
//     (8-byte loads reach only ~0.6x the HBM rate of 16-byte ones), padded to
//     a multiple of 4 entries with an id that points at a 0.0 in LDS; large
//     value-free layouts store GROUPS of five entries per row and step instead
//     (a 14-bit slot and four 12-bit deltas: tiled_layout.hpp, packed_slot);
//   * a row whose segment in some tile is much longer than the others' (the
//     column counts of simulate_data.py designs are heavy-tailed) is split into
//     chunks of <= T entries that sort next to rows of that length; every
//     chunk past the first owns an extra LDS accumulator that the epilogue
//     folds into the row in a fixed order (no atomics, bitwise reproducible);
//   * values are stored only when some entry differs from 1.0 (binary designs
//     of simulate_data.py:100-117 never read values; the unwired prototype
//     design_matrix/cython_matmal/binary_matmul.pyx:21-25 had the same idea).
//
// One workgroup (1024 threads, 16 waves, one per CU) owns a row panel and a
// group of column blocks: it fills the vector slice, streams the tile's ids
// with coalesced 1 KiB wave loads (the only HBM traffic that scales with
// nnz: 2 bytes per entry, 1.6 in groups), adds lane-private sums into LDS
// accumulators, and
// writes the panel once.  No atomics: every sum has a fixed order.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>

#include <hip/hip_ext.h>

#include "common.hpp"
#include "tiled_layout.hpp"

// -DBBX_TILED_INSTRUMENT=1 compiles the per-wave phase timers in (they cost
// ~8 SGPRs, which the production kernel has no room for); BBX_TILED_DEBUG=N
// then prints the breakdown of launches N and N+1.
#ifndef BBX_TILED_INSTRUMENT
#define BBX_TILED_INSTRUMENT 0
#endif
// The experiment arguments of tiled_spmv_kernel (`ablate`: timing-only removal
// of the gathers / slice loads / barriers; `dbg`: per-wave phase stamps) exist
// in the instrumented build ONLY: the shipped kernel has neither the
// arguments nor a branch on them.
#if BBX_TILED_INSTRUMENT
#define BBX_INSTR_PARAMS int ablate, unsigned long long* dbg,
#define BBX_INSTR_ARGS ablate, dbg,
#else
#define BBX_INSTR_PARAMS
#define BBX_INSTR_ARGS
#endif
// Register ring of the value-free kernel: RING slots of BATCH steps each.
// With non-temporal id loads the kernel is insensitive to the bytes a wave has
// in flight (even ONE 1 KiB step in flight per wave runs as fast); shallow and
// fine-grained wins by a little.  Measured at 1M x 50k, X v / X^T w in us:
// 4x2: 48.6 / 50.8   3x2: 47.6 / 50.0   4x1: 46.8 / 49.5   3x1: 46.7 / 49.3
// 2x1: 46.7 / 49.7   6x1: 47.4 / 50.1   (100k x 10k Gibbs: 4x2 614, 3x1 675 it/s)
// Value-free single-chain steps re-arm their ring slot between the LDS gathers
// and the additions (the ids are dead once the addresses exist): X~ v 43.4 ->
// 42.7 us, X~^T w 45.8 -> 45.5 us at 1M x 50k; 0 restores gather, add, re-arm.
#ifndef BBX_EARLY_ISSUE
#define BBX_EARLY_ISSUE 1
#endif
#ifndef BBX_RING_BIN
#define BBX_RING_BIN 3
#endif

namespace bbx {

// Geometry constants, BatchDesc / FoldDesc / SliceMeta and the host-side
// builder live in tiled_layout.hpp / tiled_layout.cpp (plain C++, CPU-testable).
#ifndef BBX_TILE_MIN_WAVES
#define BBX_TILE_MIN_WAVES 4  // waves per SIMD the register budget must allow
#endif
constexpr int WAVE_CHECK = LANES == WAVE ? 1 : -1;
static_assert(WAVE_CHECK == 1, "tiled layout is built for 64-lane wavefronts");

// One orientation (X or X^T) in tiled form, device resident.
struct TiledMatrix {
  int64_t R = 0, C = 0, nnz = 0;
  int W = 0, n_block = 0, PR = 0, n_panel = 0, G = 0;
  int K = 1;  // right-hand sides the geometry was sized for
  bool has_vals = false;
  bool packed = false;  // steps hold one 5-entry group per row (tiled_layout.hpp)
  int Wl = 0;           // slots of one vector slice in LDS (W, or packed_slots(W))
  int64_t n_slice = 0, n_quad = 0, n_tile = 0;
  DevMem ids;        // uint4[n_quad * 64]
  DevMem vals;       // double[n_quad * 64 * 8] when has_vals
  DevMem descs;      // BatchDesc[n_desc]: per-wave schedules
  DevMem wave_desc;  // int32[n_panel * G * 16]: first descriptor of each wave
  DevMem wg_quad0;   // uint32[n_panel * G]: first step of each workgroup's stretch
  int desc_stride = 0;  // > 0: wave k's schedule starts at k * desc_stride
  int64_t n_desc = 0;
  DevMem rowids;     // uint32[n_slice * 64]: panel-local rows A | B << 16
  DevMem folds;      // FoldDesc[n_fold]
  DevMem panel_fold; // int32[n_panel + 1]
  int n_extra = 0;   // extra accumulators per panel (row splitting)
  int split_T = 0;   // smallest split threshold used by any panel (0 = none)
  DevMem slab;       // double[G * R * K] partial sums when G > 1 (or Tdot)
  int64_t stream_bytes() const {
    return (int64_t)n_quad * 64 * 16 * (has_vals ? 5 : 1) +
           (int64_t)n_slice * 256 + (int64_t)n_desc * (int64_t)sizeof(BatchDesc);
  }
};


struct TiledPair {
  TiledMatrix x, xt;
};

constexpr int HYB_TDOT_CHUNKS = 256;  // row chunks of the dense block's D^T w
constexpr int HYB_FUSED_WAVE_KD = 1024; // widest dense block of the wave-per-row kernel
constexpr int HYB_FUSED_MAX_KD = 8192;  // ... of the workgroup-per-row-block kernel (= the split's cap)

// Mixed designs: X = B + D + S.
//
// OHDSI-style designs are binary covariates plus a few continuous ones, and the
// reference's own test helper builds simulate_design(n, p, binary_frac=.9)
// (tests/helper.py:13).  One stored value that is not 1.0 used to move the
// WHOLE matrix to the valued layout (10 bytes per entry instead of 2.4: 4x the
// time per product).  Instead the entries are split by value at construction:
//   B  every entry equal to 1.0            value-free tiled layout
//   D  the other entries of columns that hold many of them (>= n / 2): a
//      dense column-major block n x kd in f64 (a continuous covariate is a
//      dense column)
//   S  what is left                        valued tiled layout (often empty)
// All three keep the row and column numbering of X, so no vector is permuted:
//   X~ v   = epilogue(c + B v + [D v + S v])   the bracket is an n-vector the
//            value-free kernel's epilogue adds (`addend`),
//   X~^T w = the slabs of B^T w and S^T w and one more slab row holding D^T w,
//            added in this order by the common epilogue kernel.
struct HybridParts {
  TiledPair ones;        // B
  TiledPair rest;        // S (rest_nnz == 0: not built)
  int64_t ones_nnz = 0, rest_nnz = 0, dense_nnz = 0;
  int kd = 0;            // columns of D
  DevMem dense_cols;     // int32[kd]: column of X (0-based, without intercept)
  DevMem D;              // double[kd][n], column-major
  // Row-major copy double[n][ld_rm] (ld_rm = kd rounded up to 2) for the single
  // pass of an operator application, D^T (Omega (a + D v_D)) -- built for
  // DENSE_EPI_MAX < kd <= HYB_FUSED_MAX_KD (hyb_dense_fused_kernel)
  DevMem D_rm;
  int ld_rm = 0;
  bool fused_attr_set = false;  // the widest one-pass kernels' LDS attribute (per design, hence per device)
  DevMem addend;         // double[n]
  DevMem d_part;         // double[HYB_TDOT_CHUNKS][kd]: partial sums of D^T w
  // operator applications (bbx_design::in_operator): D^T (Omega t) partials
  // left by the X~ v kernel's epilogue, one row of kd per workgroup
  DevMem dw_part;        // double[NPART][kd]
  const double* dw_for = nullptr;  // the t they belong to (consumed by the next Tdot)
  uint64_t dw_serial = 0;          // ... of this operator application (bbx_design::operator_serial)
  int dw_chunks = 0;
  DevMem slab;           // double[(G_B + G_S + 1)][p]
  int n_slab = 0;
  // batches (K = 2, 4 right-hand sides) of a design WITHOUT a valued rest keep
  // the split: B in its K-layout (bbx_design::tiled_k), D shared
  bool split_k[2] = {false, false};   // slot K == 2, K == 4
  TiledPair* rest_k = nullptr;        // S in its K = 2 layout (pairs only), or null
  ~HybridParts() { delete rest_k; }
  DevMem addend_k;       // double[n][K]
  DevMem d_part_k;       // double[HYB_TDOT_CHUNKS][kd][K]
  DevMem slab_k[2];      // double[(G_B_k + 1)][p][K]
};

// ------------------------------------------------------------------ kernel

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// One step of one lane: 4 entries of its row A (e.x, e.y) and 4 of its row B
// (e.z, e.w), gathered from the vector slice in LDS.
template <bool VALS>
__device__ __forceinline__ void step_accumulate(const double* __restrict__ xs,
                                                v4u e, const v2d* v,
                                                double& a0, double& a1,
                                                double& b0, double& b1) {
  if (VALS) {
    a0 += v[0].x * xs[e.x & 0xFFFFu] + v[1].x * xs[e.y & 0xFFFFu];
    a1 += v[0].y * xs[e.x >> 16] + v[1].y * xs[e.y >> 16];
    b0 += v[2].x * xs[e.z & 0xFFFFu] + v[3].x * xs[e.w & 0xFFFFu];
    b1 += v[2].y * xs[e.z >> 16] + v[3].y * xs[e.w >> 16];
  } else {
    a0 += xs[e.x & 0xFFFFu] + xs[e.y & 0xFFFFu];
    a1 += xs[e.x >> 16] + xs[e.y >> 16];
    b0 += xs[e.z & 0xFFFFu] + xs[e.w & 0xFFFFu];
    b1 += xs[e.z >> 16] + xs[e.w >> 16];
  }
}

// The packed step (tiled_layout.hpp, packed_slot): e.x | e.y << 32 is row A's
// group, e.z | e.w << 32 row B's -- the slice slot of the first entry in bits
// 0-13, then four 12-bit forward deltas.  All five slots are gathered always
// (short groups end on a zero slot of the slice), so the decode is shifts and
// adds only: 11 VALU operations for five entries, against 8 for four plain ids.
__device__ __forceinline__ unsigned lshl3_add(unsigned d, unsigned o) {
  // o + 8 d in one operation; written in C the compiler turns the field
  // extraction and this into shift, and, add (15 operations per group for 11)
  unsigned r;
  asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(r) : "v"(d), "v"(o));
  return r;
}
typedef __attribute__((address_space(3))) const double lds_cdouble;
__device__ __forceinline__ unsigned lds_address(const double* p) {
  return (unsigned)(uintptr_t)(lds_cdouble*)p;  // p points into LDS
}
__device__ __forceinline__ double lds_read(unsigned addr) {
  return *(lds_cdouble*)(uintptr_t)addr;
}
// The five slots of a group, gathered but not yet added (the caller may issue
// the ring slot's next stream load between the gathers and the additions).
// xs_addr = lds_address(xs): the slice's LDS byte address, added once per group
// instead of once per gather.
__device__ __forceinline__ void packed_gather(unsigned xs_addr, unsigned lo,
                                              unsigned hi, double* g) {
  // byte addresses of the five slots: o_{k+1} = o_k + 8 d_k
  const unsigned o0 = lshl3_add(lo & 0x3FFFu, xs_addr);
  const unsigned o1 = lshl3_add(__builtin_amdgcn_ubfe(lo, 14, 12), o0);
  const unsigned o2 =
      lshl3_add(__builtin_amdgcn_alignbit(hi, lo, 26) & 0xFFFu, o1);
  const unsigned o3 = lshl3_add(__builtin_amdgcn_ubfe(hi, 6, 12), o2);
  const unsigned o4 = lshl3_add(hi >> 18, o3);  // bits 62-63 of a group are 0
  g[0] = lds_read(o0);
  g[1] = lds_read(o1);
  g[2] = lds_read(o2);
  g[3] = lds_read(o3);
  g[4] = lds_read(o4);
}
__device__ __forceinline__ void packed_group(unsigned xs_addr, unsigned lo,
                                             unsigned hi, double& s0,
                                             double& s1) {
  double g[5];
  packed_gather(xs_addr, lo, hi, g);
  s0 += (g[0] + g[2]) + g[4];
  s1 += g[1] + g[3];
}
// Row ids of a ring slot, split BEFORE the slot is re-armed.  As asm volatile
// statements these stay between the slot's wait and its next load; a plain
// `rr = rid[k]` kept across the re-arm makes the register allocator copy the
// (tied) wait operand -- a register with a load still in flight -- in front of
// the wait: stale row ids, sums flushed into the wrong accumulators (seen:
// v_mov_b32 v38, v54 one instruction above s_waitcnt vmcnt(4)).
__device__ __forceinline__ void split_row_ids(unsigned rr, unsigned& ra,
                                              unsigned& rb) {
  asm volatile("v_and_b32 %0, 0xffff, %2\n\tv_lshrrev_b32 %1, 16, %2"
               : "=&v"(ra), "=v"(rb)
               : "v"(rr));
}

__device__ __forceinline__ void step_accumulate_packed(
    unsigned xs_addr, v4u e, double& a0, double& a1, double& b0, double& b1) {
  packed_group(xs_addr, e.x, e.y, a0, a1);
  packed_group(xs_addr, e.z, e.w, b0, b1);
}

// The same step for K = 2 KP right-hand sides (batched chains).  The vector
// slices sit in LDS as KP planes of interleaved PAIRS, xs2[q][j] = {x_2q[j],
// x_2q+1[j]}: one ds_read_b128 serves two chains (4 LDS cycles for 64 lanes x
// 16 bytes, the byte rate of the ds_read_b64 of the single-chain kernel), and
// planes instead of a 32-byte interleave keep the 16 lanes of a b128 group
// spread over all sixteen 16-byte slots of the bank row.  Every column sees
// exactly the additions, in the order, of step_accumulate.
template <bool VALS, int KP>
__device__ __forceinline__ void step_accumulate_k(const v2d* __restrict__ xs2,
                                                  int plane, v4u e,
                                                  const v2d* v, v2d* A0,
                                                  v2d* A1, v2d* B0, v2d* B1) {
#pragma unroll
  for (int q = 0; q < KP; ++q) {
    const v2d* xp = xs2 + q * plane;
    if (VALS) {
      A0[q] += v[0].x * xp[e.x & 0xFFFFu] + v[1].x * xp[e.y & 0xFFFFu];
      A1[q] += v[0].y * xp[e.x >> 16] + v[1].y * xp[e.y >> 16];
      B0[q] += v[2].x * xp[e.z & 0xFFFFu] + v[3].x * xp[e.w & 0xFFFFu];
      B1[q] += v[2].y * xp[e.z >> 16] + v[3].y * xp[e.w >> 16];
    } else {
      A0[q] += xp[e.x & 0xFFFFu] + xp[e.y & 0xFFFFu];
      A1[q] += xp[e.x >> 16] + xp[e.y >> 16];
      B0[q] += xp[e.z & 0xFFFFu] + xp[e.w & 0xFFFFu];
      B1[q] += xp[e.z >> 16] + xp[e.w >> 16];
    }
  }
}

// The id/value/row-id stream loads are issued through inline asm so that
// hipcc does not count them: with compiler-visible loads it drains the whole
// register ring with `s_waitcnt vmcnt(0)` at every loop join (checked in the
// .s), which serialises HBM latency with the LDS gathers.  The waits are
// counted by hand instead (cdna_hip_programming.md 5.7, form (ii)): every ISSUE
// step queues exactly LOADS_PER_STEP vector-memory operations, so before
// consuming a ring slot at most (RING-1)*LOADS_PER_STEP younger ones may still
// be in flight.  Unknown extra compiler loads can only make the wait stricter.
// The id stream is read exactly once per launch: it is loaded NON-TEMPORALLY
// (`nt`), so that it does not evict what the kernel re-reads -- the vector
// slices every workgroup refills, the schedules -- from L2 and the Infinity
// Cache.  Measured at 1M x 50k: 53.0 -> 48.6 us (X v), 55.3 -> 51.1 us (X^T w);
// `sc1` / `sc0 sc1` make no difference.  The same hint on the row-id loads
// costs 1.3 us and on the value loads of the valued kernel 50 % (0.186 ->
// 0.282 ms), so those stay temporal.
#ifndef BBX_IDS_MOD
#define BBX_IDS_MOD " nt"
#endif
#ifndef BBX_VALS_MOD
#define BBX_VALS_MOD ""
#endif
#ifndef BBX_RID_MOD
#define BBX_RID_MOD ""
#endif
__device__ __forceinline__ void asm_load_x4(v4u& dst, unsigned off,
                                            const void* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2" BBX_IDS_MOD
               : "=v"(dst)
               : "v"(off), "s"(base)
               : "memory");
}
__device__ __forceinline__ void asm_load_d2(v2d& dst, unsigned off,
                                            const void* base) {
  asm volatile("global_load_dwordx4 %0, %1, %2" BBX_VALS_MOD
               : "=v"(dst)
               : "v"(off), "s"(base)
               : "memory");
}
__device__ __forceinline__ void asm_load_u32(unsigned& dst, unsigned off,
                                             const void* base) {
  asm volatile("global_load_dword %0, %1, %2" BBX_RID_MOD
               : "=v"(dst)
               : "v"(off), "s"(base)
               : "memory");
}

constexpr int FILL_UNROLL = (TILE_W_MAX + TILE_THREADS - 1) / TILE_THREADS;
// batched kernels: K slices share the LDS, K * W <= (158 KB / 8) doubles
constexpr int FILL_K_PAIRS =
    ((TILE_LDS_BYTES - 2048) / 16 + TILE_THREADS - 1) / TILE_THREADS;

// KP == 0: one right-hand side (the single-chain kernel).  KP > 0: K = 2 KP
// right-hand sides share the pass over the id stream; x is the interleaved
// [C][K] input, x0_ptr its K intercept entries, c_part / out_sum_part hold one
// NPART-block per chain `part_stride` doubles apart, rowscale_k / out_k are
// per-chain pointers (out_k.p[c][row * out_stride]: out_stride = K with
// p[c] = base + c writes an interleaved [R][K] result), slab is [G][R][K].
//
// FOLD (KP == 0, WIDE): the direction step of CG iteration fa.k rides in this
// launch (common.hpp DotFold): x is s.*r (ONE vector: the slices do not depend
// on beta), the epilogue forms t_k = X~(s.*r_k) + beta t_{k-1} (the product is
// linear in its input), and the workgroup writes p and <p, d p> for its share
// of the coordinates.
//
// DENSEP (KP == 0, direct epilogue): the dense block of a mixed design rides in
// the epilogue (common.hpp DenseEpi).
template <bool VALS, bool WIDE, int KP, bool FOLD = false, bool DENSEP = false,
          bool PACK = false>
__global__ __launch_bounds__(TILE_THREADS, BBX_TILE_MIN_WAVES) void tiled_spmv_kernel(
    int64_t R, int64_t C, int W, int PR, int G, int blocks_per_group,
    const int32_t* __restrict__ wave_desc, int desc_stride,
    const BatchDesc* __restrict__ descs, const uint32_t* __restrict__ rowids,
    const uint32_t* __restrict__ wg_quad0,
    const uint4* __restrict__ ids_all, const double* __restrict__ vals_all,
    const double* __restrict__ x,
    // epilogue (direct mode, G == 1 and out != nullptr):
    //   out[r] = rowscale[r] * (c0 - sum(c_part) + acc)
    const double* __restrict__ c_part, const double* x0_ptr,
    const double* __restrict__ rowscale, double* __restrict__ out,
    double* __restrict__ slab, int n_acc,
    const int32_t* __restrict__ panel_fold, const FoldDesc* __restrict__ folds,
    double* __restrict__ out_sum_part, int twt_off,
    BBX_INSTR_PARAMS const int* __restrict__ skip_flag,
    ChainPtrs rowscale_k, ChainOut out_k, int out_stride, int part_stride,
    const double* __restrict__ addend, DotFold fa, DenseEpi de) {
  static_assert(!FOLD || (KP == 0 && WIDE), "the folded direction step is single-chain");
  static_assert(!DENSEP || (KP == 0 && !FOLD), "dense epilogue: single chain, plain loop");
  static_assert(!PACK || (KP == 0 && !VALS), "packed groups: value-free, one right-hand side");
  constexpr int K = KP > 0 ? 2 * KP : 1;
#if !BBX_TILED_INSTRUMENT
  constexpr int ablate = 0;                      // (compile-time: every
  constexpr unsigned long long* dbg = nullptr;   // `if (dbg)` below folds away)
#endif
  // (scalar load, issued first; checked below once the descriptor loads that
  // every launch needs anyway have been issued, so it adds no round trip)
  const int skip = skip_flag ? *skip_flag : 0;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // slice slots: one per column, plus (PACK) a zero slot after every 4095
  // columns; xs[Wl .. Wl + 8) == 0 (padding target / terminal zero slot)
  const int Wl = PACK ? W + (W - 1) / PACK_PERIOD : W;
  double* xs = lds;               // Wl + 8 doubles
  const unsigned xs_addr = lds_address(xs);
  double* acc = lds + (Wl + 8);   // n_acc = PR + extra doubles
  // batched: KP planes of (W + 8) pairs, then KP planes of n_acc pairs
  v2d* xs2 = reinterpret_cast<v2d*>(lds);
  v2d* acc2 = xs2 + KP * (W + 8);
  const int xplane = W + 8;
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  // Placement: workgroups go to the 8 XCDs round robin (blockIdx.x % 8), and the
  // panels of one column-block group all load the same vector slices, so they
  // should share an L2.  With G a multiple of 8 the plain numbering (panel
  // major) does that by itself; the K = 2 layout of X^T at 1M x 50k has G = 15
  // and every XCD fetched every slice (PMC: 401 MB per launch against 287 MB
  // algorithmic).  So XCD x takes a CONTIGUOUS range of the group-major order;
  // `bid` = panel * G + group stays the workgroup's logical number (schedules,
  // partial-sum slots: results do not depend on the placement).
  const int n_wg = (int)gridDim.x;
  const int xcd_q = n_wg >> 3, xcd_r = n_wg & 7;
  const int xcd = (int)blockIdx.x & 7;
  const int gm = xcd * xcd_q + (xcd < xcd_r ? xcd : xcd_r) + ((int)blockIdx.x >> 3);
  const int n_panel_wg = n_wg / G;
  const int group = gm / n_panel_wg;
  const int panel = gm - group * n_panel_wg;
  const int bid = panel * G + group;
  const int64_t row0 = (int64_t)panel * PR;
  const int rows_here = (int)((R - row0 < PR) ? (R - row0) : PR);
  // This workgroup's stretch of the id (and value) stream: a 64-bit base, the
  // schedules' steps count from it and every byte offset below is 32-bit --
  // the stream as a whole may exceed 4 GiB (a valued design of 6e8 entries: 38 GB)
  const size_t wg_q0 = (size_t)wg_quad0[bid];
  const uint4* __restrict__ ids = ids_all + wg_q0 * WAVE;
  const double* __restrict__ vals = VALS ? vals_all + wg_q0 * WAVE * 8 : nullptr;

  if constexpr (KP > 0) {
    const v2d zero2 = {0., 0.};
    for (int r = tid; r < KP * n_acc; r += TILE_THREADS) acc2[r] = zero2;
    if (tid < 8 * KP) xs2[(tid >> 3) * xplane + W + (tid & 7)] = zero2;
  } else {
    for (int r = tid; r < n_acc; r += TILE_THREADS) acc[r] = 0.;
    if (tid < 8) xs[Wl + tid] = 0.;
    if constexpr (PACK) {
      // the slice fills never touch the zero slots: written once
      const int z = (tid - 8) * (PACK_PERIOD + 1) + PACK_PERIOD;
      if (tid >= 8 && z < Wl) xs[z] = 0.;
    }
  }

  constexpr int BATCH = VALS ? BATCH_VAL : BATCH_BIN;  // steps per ring slot
  constexpr int RING = VALS ? 2 : BBX_RING_BIN;  // slots: RING-1 batches in flight
  constexpr int NV = VALS ? 4 : 1;     // 16-byte value loads per step
  constexpr int LOADS_PER_STEP = BATCH * (VALS ? 5 : 1) + 1;
  constexpr int WAIT_COUNT = (RING - 1) * LOADS_PER_STEP;
  static_assert(WAIT_COUNT < 64, "vmcnt is a 6-bit field");

  // Each wave walks its own precomputed schedule (BatchDesc stream) through a
  // ring of RING register slots: while one batch is gathered from LDS the next
  // RING-1 are in flight from HBM, across slice AND tile boundaries.  The
  // schedule marks where the wave enters a new tile; there every wave of the
  // workgroup meets at a barrier and the vector slice in LDS is replaced.
  // Column block of the tile being processed.  A workgroup's tiles are the
  // consecutive column blocks of its group (empty ones included), so the block
  // index is arithmetic: no descriptor load sits between reaching a tile
  // boundary and issuing the loads of the next vector slice.
  int cb = group * blocks_per_group - 1;  // advanced at every switch
  // Issue cursor.  The wave's descriptors are fetched 64 at a time (one per
  // lane) and read back with v_readlane, so that no memory latency sits
  // between two ISSUE steps; the next block of 64 is prefetched.
  // With equal-stride schedules the first descriptor block needs no lookup
  // (one dependent memory round trip less before the first stream load).
  int blk = desc_stride > 0
                ? (int)(bid * TILE_WAVES + wave) * desc_stride
                : wave_desc[bid * TILE_WAVES + wave];
  int pos = 0;
  const uint4* __restrict__ desc4 = reinterpret_cast<const uint4*>(descs);
  uint4 dcur = desc4[blk + lane];
  uint4 dnxt = desc4[blk + WAVE + lane];
  // FOLD: everything the direction step reads goes out with the descriptor
  // loads -- one round trip for both.  Every WAVE re-adds the partials itself
  // (2 KB from L2 per wave): no LDS broadcast, no barrier.
  double f_rr[NPART / WAVE], f_cr[NPART / WAVE];
  double f_atol = 0., f_rho_prev = 1., f_x0r = 0.;
  double f_r0 = 0., f_p0 = 0., f_d0 = 0.;
  double f_beta = 0., f_c = 0., f_pdp = 0.;
  int64_t f_j0 = 0, f_j1 = 0;
  if constexpr (FOLD) {
#pragma unroll
    for (int k4 = 0; k4 < NPART / WAVE; ++k4) {
      f_rr[k4] = fa.rr_part[lane + k4 * WAVE];
      f_cr[k4] = fa.cr_part[lane + k4 * WAVE];
    }
    f_atol = fa.st->atol;
    if (fa.k > 0) f_rho_prev = fa.st->rho[(fa.k - 1) & 1];
    if (fa.intercept) f_x0r = fa.sr[0];
    // this workgroup's share of the P coordinates (p and <p, d p>)
    const int64_t chunk = (fa.P + n_wg - 1) / n_wg;
    f_j0 = (int64_t)bid * chunk;
    f_j1 = f_j0 + chunk < fa.P ? f_j0 + chunk : fa.P;
    if (f_j0 + tid < f_j1) {
      const int64_t j = f_j0 + tid;
      f_r0 = fa.r[j];
      f_d0 = fa.d[j];
      if (fa.k > 0) f_p0 = fa.pvec[j];
    }
  }
  // Retire every compiler-visible load before the ring starts (see ISSUE).
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
  if (skip) return;  // the CG solve this launch belongs to has already stopped
  if constexpr (FOLD) {
    // top of SciPy's cg loop (vecops.hip cg_direction_kernel): rho = r.r, the
    // same adds in the same order in every wave of every workgroup
    double a = 0., cr = 0.;
#pragma unroll
    for (int k4 = 0; k4 < NPART / WAVE; ++k4) {
      a += f_rr[k4];
      cr += f_cr[k4];
    }
    const double rho = wave_allsum(a);
    cr = wave_allsum(cr);
    const bool finite = (rho == rho) && (rho - rho == 0.);
    if (!finite || sqrt(rho) < f_atol) {   // uniform over the whole grid
      if (bid == 0 && tid == 0) {
        fa.st->done = 1;
        fa.st->running = 0;
        if (!finite) fa.st->bad = 1;
        if (fa.word)
          cg_word_store(fa.word, fa.tag | CG_WORD_DONE |
                                     (finite ? 0ull : CG_WORD_BAD) |
                                     (unsigned long long)fa.k);
      }
      return;
    }
    if (fa.word && bid == 0 && tid == 0)
      cg_word_store(fa.word, fa.tag | (unsigned long long)(fa.k + 1));
    const double beta = fa.k > 0 ? rho / f_rho_prev : 0.;
    // wave-uniform scalars; c = v0 - <offset, v[1:]> of v = s.*r
    f_beta = lane_value(beta, 0);
    f_c = lane_value(f_x0r - cr, 0);
    if (bid == 0 && tid == 0) fa.st->rho[fa.k & 1] = rho;
    for (int64_t j = f_j0 + tid; j < f_j1; j += TILE_THREADS) {
      const bool first = j == f_j0 + tid;
      double pj = first ? f_r0 : fa.r[j];
      if (fa.k > 0) pj = fma(f_beta, first ? f_p0 : fa.pvec[j], pj);
      fa.pvec[j] = pj;
      f_pdp = fma((first ? f_d0 : fa.d[j]) * pj, pj, f_pdp);
    }
  }
  // (Round 2 tried to move the FIRST slice fill up here, next to the descriptor
  // loads, and to enter the loop with the ring primed but not waited for: the
  // kernel then faulted intermittently.  Without the vmcnt(0) of the first
  // tile switch the compiler's loop-entry copies of the ring registers can run
  // before the asm loads have landed; a landing load then overwrites a
  // register the compiler has re-used, e.g. for an address.  With a vmcnt(0)
  // after the priming the start-up is two dependent round trips again --
  // descriptors, then ids -- exactly as below, so nothing is gained.)
  v4u e[RING][BATCH];
  v2d ev[RING][BATCH][NV];
  unsigned rid[RING];
  unsigned info[RING];
  double a0 = 0., a1 = 0., b0 = 0., b1 = 0.;
  constexpr int KPA = KP > 0 ? KP : 1;
  v2d A0[KPA], A1[KPA], B0[KPA], B1[KPA];
#pragma unroll
  for (int q = 0; q < KPA; ++q) {
    A0[q] = v2d{0., 0.};
    A1[q] = v2d{0., 0.};
    B0[q] = v2d{0., 0.};
    B1[q] = v2d{0., 0.};
  }
  // BBX_TILED_DEBUG (instrumented build): per-wave cycle stamps (start, time
  // in tile switches, end of the stream loop, end of the kernel).
  // 32-bit tick counts (durations only: s_memtime bases differ across XCDs)
  unsigned t_start = 0, t_switch = 0, t_loop = 0, t_skew = 0, t_drain = 0;
  if (dbg) t_start = (unsigned)__builtin_amdgcn_s_memtime();

#define BBX_ISSUE(K)                                                          \
  do {                                                                        \
    const unsigned d_quad0 =                                                  \
        (unsigned)__builtin_amdgcn_readlane((int)dcur.x, pos);                \
    const unsigned d_row =                                                    \
        (unsigned)__builtin_amdgcn_readlane((int)dcur.y, pos);                \
    const unsigned inf =                                                      \
        (unsigned)__builtin_amdgcn_readlane((int)dcur.z, pos);                \
    const int cntk = (int)(inf & 15u);                                        \
    _Pragma("unroll") for (int u = 0; u < BATCH; ++u) {                       \
      /* Steps past the end of a slice (and the loads of marker batches)   */ \
      /* only keep the vmcnt bookkeeping uniform.  They must NOT re-read a */ \
      /* line that is still in flight (the L1 parks such a request until   */ \
      /* the line lands and blocks every request behind it): all lanes     */ \
      /* read the first 16 bytes of the stream, one resident line.         */ \
      const unsigned slot =                                                   \
          (u < cntk) ? (d_quad0 + (unsigned)u) * WAVE + lane : 0u;            \
      asm_load_x4(e[K][u], slot * 16u, ids);                                  \
      if (VALS) {                                                             \
        _Pragma("unroll") for (int j = 0; j < NV; ++j)                        \
            asm_load_d2(ev[K][u][j], slot * 64u + 16u * j, vals);             \
      }                                                                       \
    }                                                                         \
    asm_load_u32(rid[K], (cntk > 0 ? (d_row + lane) : 0u) * 4u, rowids);      \
    info[K] = inf;                                                            \
    if (!(inf & BD_END)) {                                                    \
      ++pos;                                                                  \
      if (pos == WAVE) {                                                      \
        pos = 0;                                                              \
        blk += WAVE;                                                          \
        dcur = dnxt;                                                          \
        dnxt = desc4[blk + WAVE + lane];                                      \
        /* retire this (compiler-visible) load here, once per 64 batches: */  \
        /* left pending it makes hipcc guard every later register write   */  \
        /* in the loop with vmcnt(0), which drains the ring (seen in .s)  */  \
        __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0) only */               \
      }                                                                       \
    }                                                                         \
  } while (0)

// Wait until slot K's loads have landed (all later ISSUE steps may still be
// in flight) and make its registers opaque to the scheduler at this point.
#define BBX_WAIT(K)                                                           \
  do {                                                                        \
    if (VALS) {                                                               \
      asm volatile("s_waitcnt vmcnt(%6)"                                      \
                   : "+v"(e[K][0]), "+v"(rid[K]), "+v"(ev[K][0][0]),          \
                     "+v"(ev[K][0][1]), "+v"(ev[K][0][2]), "+v"(ev[K][0][3])  \
                   : "n"(WAIT_COUNT)                                          \
                   : "memory");                                               \
    } else {                                                                  \
      asm volatile("s_waitcnt vmcnt(%2)"                                      \
                   : "+v"(e[K][0]), "+v"(rid[K])                              \
                   : "n"(WAIT_COUNT)                                          \
                   : "memory");                                               \
      /* every other register of the slot is tied to a statement AFTER the */ \
      /* wait (asm volatile statements keep their order), so no use of it  */ \
      /* can be scheduled above the wait; each operand appears once        */ \
      _Pragma("unroll") for (int u_ = 1; u_ < BATCH; ++u_)                    \
          asm volatile("" : "+v"(e[K][u_]));                                  \
    }                                                                         \
  } while (0)

#pragma unroll
  for (int k = 0; k < RING; ++k) BBX_ISSUE(k);
  bool done = false;
  while (!done) {
#pragma unroll
    for (int k = 0; k < RING; ++k) {
      if (!done) {
        const unsigned inf = info[k];
        if (inf & BD_END) {
          done = true;
        } else {
          if (inf & BD_TILE_FIRST) {
            // ---- enter the next tile: replace the vector slice in LDS
            unsigned t_sw0 = 0;
            if (dbg) t_sw0 = (unsigned)__builtin_amdgcn_s_memtime();
            ++cb;
            const int64_t col0 = (int64_t)cb * W;
            const int cols_here = (int)((C - col0 < W) ? (C - col0) : W);
            // The slice moves through the CU's L1 at 64 B/clk (100-127 KB per
            // switch, ~1 us): 16-byte lane loads where the source is 16-byte
            // aligned (8-byte accesses reach ~0.6x the rate), pairs of doubles
            // per thread, otherwise one double per lane and load.
            if constexpr (KP > 0) {
              // batched: the interleaved input is 16-byte aligned by
              // construction; element m = j * KP + q of the slice goes to
              // plane q, slot j
              v2d fk[FILL_K_PAIRS];
              const v2d* gx = reinterpret_cast<const v2d*>(x) + col0 * KP;
              const int m_here = cols_here * KP, m_all = W * KP;
#pragma unroll
              for (int u = 0; u < FILL_K_PAIRS; ++u) {
                const int m = tid + u * TILE_THREADS;
                if (m < m_here && !(ablate & 2)) fk[u] = gx[m];
                else fk[u] = v2d{0., 0.};
              }
              if (!(ablate & 4)) __syncthreads();
              if (dbg) t_skew += (unsigned)__builtin_amdgcn_s_memtime() - t_sw0;
              __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
#pragma unroll
              for (int u = 0; u < FILL_K_PAIRS; ++u) {
                const int m = tid + u * TILE_THREADS;
                if (m < m_all) xs2[(m % KP) * xplane + m / KP] = fk[u];
              }
            } else {
              constexpr int FILL_PAIRS = (FILL_UNROLL + 1) / 2;
              v2d fp[FILL_PAIRS];
              // (WIDE is chosen by the launcher: x and W * 8 are 16-byte aligned)
              constexpr bool wide = WIDE;
              if (wide) {
  #pragma unroll
                for (int u = 0; u < FILL_PAIRS; ++u) {
                  const int j = 2 * (tid + u * TILE_THREADS);
                  if (j + 1 < cols_here && !(ablate & 2)) {
                    fp[u] = *reinterpret_cast<const v2d*>(x + col0 + j);
                  } else {
                    fp[u].x = (j < cols_here && !(ablate & 2)) ? x[col0 + j] : 0.;
                    fp[u].y = 0.;
                  }
                }
              } else {
  #pragma unroll
                for (int u = 0; u < FILL_PAIRS; ++u) {
                  const int j0 = tid + (2 * u) * TILE_THREADS;
                  const int j1 = tid + (2 * u + 1) * TILE_THREADS;
                  fp[u].x = (j0 < cols_here && !(ablate & 2)) ? x[col0 + j0] : 0.;
                  fp[u].y = (j1 < cols_here && !(ablate & 2)) ? x[col0 + j1] : 0.;
                }
              }
              if (dbg) {
                // (instrumented builds: how much of the wait below is this
                // wave's OWN outstanding loads -- the slice values just asked
                // for and the ring's younger steps, which the barrier's fence
                // retires anyway -- and how much the other waves)
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
                t_drain += (unsigned)__builtin_amdgcn_s_memtime() - t_sw0;
              }
              if (!(ablate & 4))
                __syncthreads();  // every wave is done with the previous slice
              if (dbg) t_skew += (unsigned)__builtin_amdgcn_s_memtime() - t_sw0;
              // one explicit wait for the slice values on every path, so that no
              // compiler-visible load is left "maybe pending" inside the loop
              __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
              if (wide) {
  #pragma unroll
                for (int u = 0; u < FILL_PAIRS; ++u) {
                  const int j = 2 * (tid + u * TILE_THREADS);
                  if constexpr (PACK) {
                    // (a pair may straddle a zero slot: two 8-byte stores)
                    if (j < W) {
                      xs[j + j / PACK_PERIOD] = fp[u].x;
                      xs[j + 1 + (j + 1) / PACK_PERIOD] = fp[u].y;  // W even
                    }
                  } else {
                    if (j < W) *reinterpret_cast<v2d*>(xs + j) = fp[u];  // W even
                  }
                }
              } else {
  #pragma unroll
                for (int u = 0; u < FILL_PAIRS; ++u) {
                  const int j0 = tid + (2 * u) * TILE_THREADS;
                  const int j1 = tid + (2 * u + 1) * TILE_THREADS;
                  if (j0 < W) xs[PACK ? j0 + j0 / PACK_PERIOD : j0] = fp[u].x;
                  if (j1 < W) xs[PACK ? j1 + j1 / PACK_PERIOD : j1] = fp[u].y;
                }
              }
            }
            if (!(ablate & 4)) __syncthreads();
            if (dbg) t_switch += (unsigned)__builtin_amdgcn_s_memtime() - t_sw0;
          }
          const int cntk = (int)(inf & 15u);
          bool issued = false;
          if constexpr (!PACK && !VALS && KP == 0 && BATCH == 1 && BBX_EARLY_ISSUE) {
            if (cntk > 0) {
              // (plain ids: the same order -- gathers, re-arm, additions)
              BBX_WAIT(k);
              const v4u ee = e[k][0];
              const double g0 = xs[ee.x & 0xFFFFu], g1 = xs[ee.y & 0xFFFFu];
              const double g2 = xs[ee.x >> 16], g3 = xs[ee.y >> 16];
              const double g4 = xs[ee.z & 0xFFFFu], g5 = xs[ee.w & 0xFFFFu];
              const double g6 = xs[ee.z >> 16], g7 = xs[ee.w >> 16];
              unsigned ra, rb;
              split_row_ids(rid[k], ra, rb);
              BBX_ISSUE(k);
              issued = true;
              a0 += g0 + g1;
              a1 += g2 + g3;
              b0 += g4 + g5;
              b1 += g6 + g7;
              if (inf & BD_LAST) {
                if (ra != NO_ROW) acc[ra] += a0 + a1;
                if (rb != NO_ROW) acc[rb] += b0 + b1;
                a0 = a1 = b0 = b1 = 0.;
              }
            }
          } else
          if constexpr (PACK && BATCH == 1 && BBX_EARLY_ISSUE) {
            if (cntk > 0) {
              // decode, gather, then re-arm the slot BEFORE the additions: the
              // ids are dead once the addresses exist, and the next stream
              // load need not wait for the LDS round trip
              BBX_WAIT(k);
              double ga[5], gb[5];
              packed_gather(xs_addr, e[k][0].x, e[k][0].y, ga);
              packed_gather(xs_addr, e[k][0].z, e[k][0].w, gb);
              unsigned ra, rb;
              split_row_ids(rid[k], ra, rb);
              BBX_ISSUE(k);
              issued = true;
              a0 += (ga[0] + ga[2]) + ga[4];
              a1 += ga[1] + ga[3];
              b0 += (gb[0] + gb[2]) + gb[4];
              b1 += gb[1] + gb[3];
              if (inf & BD_LAST) {
                if (ra != NO_ROW) acc[ra] += a0 + a1;
                if (rb != NO_ROW) acc[rb] += b0 + b1;
                a0 = a1 = b0 = b1 = 0.;
              }
            }
          } else
          if (cntk > 0) {
            BBX_WAIT(k);
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
              if (u < cntk) {
                if (ablate & 1)
                  a0 += (double)(e[k][u].x ^ e[k][u].y ^ e[k][u].z ^ e[k][u].w);
                else if constexpr (KP > 0)
                  step_accumulate_k<VALS, KP>(xs2, xplane, e[k][u], ev[k][u],
                                              A0, A1, B0, B1);
                else if constexpr (PACK)
                  step_accumulate_packed(xs_addr, e[k][u], a0, a1, b0, b1);
                else
                  step_accumulate<VALS>(xs, e[k][u], ev[k][u], a0, a1, b0, b1);
              }
            if (inf & BD_LAST) {
              const unsigned rr = rid[k];
              const unsigned ra = rr & 0xFFFFu, rb = rr >> 16;
              if constexpr (KP > 0) {
#pragma unroll
                for (int q = 0; q < KP; ++q) {
                  if (ra != NO_ROW) acc2[q * n_acc + ra] += A0[q] + A1[q];
                  if (rb != NO_ROW) acc2[q * n_acc + rb] += B0[q] + B1[q];
                  A0[q] = v2d{0., 0.};
                  A1[q] = v2d{0., 0.};
                  B0[q] = v2d{0., 0.};
                  B1[q] = v2d{0., 0.};
                }
              } else {
                if (ra != NO_ROW) acc[ra] += a0 + a1;
                if (rb != NO_ROW) acc[rb] += b0 + b1;
                a0 = a1 = b0 = b1 = 0.;
              }
            }
          }
          if (!issued) BBX_ISSUE(k);
        }
      }
    }
  }
  // Drain the dummy loads still in flight; naming every slot keeps the
  // destination registers allocated until the data has landed.
#pragma unroll
  for (int k = 0; k < RING; ++k) {
    if (VALS) {
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(e[k][0]), "+v"(rid[k]), "+v"(ev[k][0][0]),
                     "+v"(ev[k][0][1]), "+v"(ev[k][0][2]), "+v"(ev[k][0][3])
                   :
                   : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(e[k][0]), "+v"(rid[k])
                   :
                   : "memory");
#pragma unroll
      for (int u_ = 1; u_ < BATCH; ++u_) asm volatile("" : "+v"(e[k][u_]));
    }
  }
#undef BBX_ISSUE
#undef BBX_WAIT
  if (dbg) t_loop = (unsigned)__builtin_amdgcn_s_memtime() - t_start;
  if constexpr (KP > 0) {
    // ---- batched epilogue: the single-chain one below, once per column
    constexpr int EPI_K = (TILE_PR_MAX / K) / TILE_THREADS;
    double rs_pre[EPI_K][K];
    double cp_pre[K][NPART / WAVE];
    double x0_pre[K];
    if (out_k.p[0]) {
#pragma unroll
      for (int u = 0; u < EPI_K; ++u) {
        const int r = tid + u * TILE_THREADS;
#pragma unroll
        for (int c = 0; c < K; ++c)
          rs_pre[u][c] = (rowscale_k.p[c] && r < rows_here)
                             ? rowscale_k.p[c][row0 + r] : 1.;
      }
#pragma unroll
      for (int c = 0; c < K; ++c) {
#pragma unroll
        for (int k = 0; k < NPART / WAVE; ++k)
          cp_pre[c][k] = c_part ? c_part[c * part_stride + lane + k * WAVE] : 0.;
        x0_pre[c] = x0_ptr ? x0_ptr[c] : 0.;
      }
    }
    __syncthreads();
    {  // fold the chunk accumulators of split rows, fixed order
      const int f0 = panel_fold[panel], f1 = panel_fold[panel + 1];
      for (int f = f0 + tid; f < f1; f += TILE_THREADS) {
        const FoldDesc fd = folds[f];
#pragma unroll
        for (int q = 0; q < KP; ++q) {
          v2d v = acc2[q * n_acc + fd.row];
          for (int c = 0; c < fd.count; ++c) v += acc2[q * n_acc + fd.first + c];
          acc2[q * n_acc + fd.row] = v;
        }
      }
      if (f1 > f0) __syncthreads();
    }
    double* scratch = lds;  // the slices are dead: 3 K + 2 K TILE_WAVES doubles
    if (out_k.p[0]) {
      if (tid < WAVE) {
#pragma unroll
        for (int c = 0; c < K; ++c) {
          double cs = 0.;
          if (c_part) {
#pragma unroll
            for (int k = 0; k < NPART / WAVE; ++k) cs += cp_pre[c][k];
            cs = wave_allsum(cs);
          }
          if (tid == 0) scratch[c] = x0_pre[c] - cs;
        }
      }
      __syncthreads();
      double cc[K], tsum[K], t2sum[K];
#pragma unroll
      for (int c = 0; c < K; ++c) {
        cc[c] = scratch[c];
        tsum[c] = 0.;
        t2sum[c] = 0.;
      }
#pragma unroll
      for (int u = 0; u < EPI_K; ++u) {
        const int r = tid + u * TILE_THREADS;
        if (r < rows_here) {
#pragma unroll
          for (int q = 0; q < KP; ++q) {
            const v2d a = acc2[q * n_acc + r];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const int c = 2 * q + hh;
              double t = cc[c] + (hh ? a.y : a.x);
              // mixed designs: the dense block's product, interleaved [n][K]
              if (addend) t += addend[(int64_t)(row0 + r) * K + c];
              double v = t;
              if (rowscale_k.p[c]) v *= rs_pre[u][c];
              out_k.p[c][(row0 + r) * out_stride] = v;
              tsum[c] += v;
              t2sum[c] += v * t;
            }
          }
        }
      }
      if (out_sum_part) {
#pragma unroll
        for (int c = 0; c < K; ++c) {
          tsum[c] = wave_allsum(tsum[c]);
          t2sum[c] = wave_allsum(t2sum[c]);
        }
        __syncthreads();
        if (lane == 0) {
#pragma unroll
          for (int c = 0; c < K; ++c) {
            scratch[K + (2 * c) * TILE_WAVES + wave] = tsum[c];
            scratch[K + (2 * c + 1) * TILE_WAVES + wave] = t2sum[c];
          }
        }
        __syncthreads();
        if (tid < K) {
          double tot = 0., tot2 = 0.;
          for (int wv = 0; wv < TILE_WAVES; ++wv) {
            tot += scratch[K + (2 * tid) * TILE_WAVES + wv];
            tot2 += scratch[K + (2 * tid + 1) * TILE_WAVES + wv];
          }
          out_sum_part[tid * part_stride + bid] = tot;
          if (twt_off)
            out_sum_part[tid * part_stride + twt_off + bid] = tot2;
        }
        // consumers add NPART slots: the first workgroup clears the unused ones
        if (bid == 0 && (int)gridDim.x + tid < NPART) {
#pragma unroll
          for (int c = 0; c < K; ++c) {
            out_sum_part[c * part_stride + gridDim.x + tid] = 0.;
            if (twt_off)
              out_sum_part[c * part_stride + twt_off + (int)gridDim.x + tid] = 0.;
          }
        }
      }
    } else {
      v2d* dst = reinterpret_cast<v2d*>(slab) + ((int64_t)group * R + row0) * KP;
      for (int r = tid; r < rows_here; r += TILE_THREADS) {
#pragma unroll
        for (int q = 0; q < KP; ++q) dst[(int64_t)r * KP + q] = acc2[q * n_acc + r];
      }
    }
  } else {
    // Everything the epilogue reads from memory and that does not depend on the
    // accumulators is requested NOW, before the wave joins the final barrier:
    // the round trip (the row scale is an HBM-resident n-vector) overlaps the
    // wait for the slowest wave instead of following it.
    constexpr int EPI_UNROLL = TILE_PR_MAX / TILE_THREADS;
    double rs_pre[EPI_UNROLL];
    double tu_pre[FOLD ? EPI_UNROLL : 1];   // FOLD: the previous iteration's t
    double cp_pre[NPART / WAVE];
    double x0_pre = 0.;
    if (out) {
  #pragma unroll
      for (int u = 0; u < EPI_UNROLL; ++u) {
        const int r = tid + u * TILE_THREADS;
        rs_pre[u] = (rowscale && r < rows_here) ? rowscale[row0 + r] : 1.;
        if constexpr (FOLD)
          tu_pre[u] = (fa.k > 0 && r < rows_here) ? fa.tu[row0 + r] : 0.;
      }
      if constexpr (!FOLD) {
  #pragma unroll
        for (int k = 0; k < NPART / WAVE; ++k)
          cp_pre[k] = c_part ? c_part[lane + k * WAVE] : 0.;
        if (x0_ptr) x0_pre = *x0_ptr;
      }
    }
    __syncthreads();
    {  // fold the chunk accumulators of split rows, fixed order
      const int f0 = panel_fold[panel], f1 = panel_fold[panel + 1];
      for (int f = f0 + tid; f < f1; f += TILE_THREADS) {
        const FoldDesc fd = folds[f];
        double v = acc[fd.row];
        for (int c = 0; c < fd.count; ++c) v += acc[fd.first + c];
        acc[fd.row] = v;
      }
      if (f1 > f0) __syncthreads();
    }
    if (out) {
      // direct epilogue: c = x0 - sum(c_part), summed once in a fixed order
      // (FOLD: every thread holds it since the direction step)
      double c = f_c;
      if constexpr (!FOLD) {
        if (tid < WAVE) {
          double cs = 0.;
          if (c_part) {
  #pragma unroll
            for (int k = 0; k < NPART / WAVE; ++k) cs += cp_pre[k];
            cs = wave_allsum(cs);
          }
          if (tid == 0) xs[0] = x0_pre - cs;
        }
        __syncthreads();
        c = xs[0];
      }
      double tsum = 0., t2sum = 0.;
      if constexpr (DENSEP) {
        // mixed design: t_r = c + (B v)_r + sum_j D[j][r] v_j, w_r = Omega_r t_r,
        // then this workgroup's part of D^T w (fixed order: rows of a thread,
        // lanes, waves) -- D comes from HBM once and from L2 the second time
        __shared__ double s_dw[DENSE_EPI_MAX * TILE_WAVES];
        double tt[EPI_UNROLL], ww[EPI_UNROLL];
  #pragma unroll
        for (int u = 0; u < EPI_UNROLL; ++u) {
          const int r = tid + u * TILE_THREADS;
          tt[u] = r < rows_here ? c + acc[r] : 0.;
        }
        for (int j = 0; j < de.kd; ++j) {
          const double vj = x[de.cols[j]];
          const double* __restrict__ Dj = de.D + (int64_t)j * de.n + row0;
  #pragma unroll
          for (int u = 0; u < EPI_UNROLL; ++u) {
            const int r = tid + u * TILE_THREADS;
            if (r < rows_here) tt[u] = fma(Dj[r], vj, tt[u]);
          }
        }
  #pragma unroll
        for (int u = 0; u < EPI_UNROLL; ++u) {
          const int r = tid + u * TILE_THREADS;
          ww[u] = 0.;
          if (r < rows_here) {
            double v = tt[u];
            if (rowscale) v *= rs_pre[u];
            out[row0 + r] = v;
            tsum += v;
            t2sum += v * tt[u];
            ww[u] = v;
          }
        }
        for (int j = 0; j < de.kd; ++j) {
          const double* __restrict__ Dj = de.D + (int64_t)j * de.n + row0;
          double a = 0.;
  #pragma unroll
          for (int u = 0; u < EPI_UNROLL; ++u) {
            const int r = tid + u * TILE_THREADS;
            if (r < rows_here) a = fma(Dj[r], ww[u], a);
          }
          a = wave_allsum(a);
          if (lane == 0) s_dw[j * TILE_WAVES + wave] = a;
        }
        __syncthreads();
        if (tid < de.kd) {
          double tot = 0.;
          for (int wv = 0; wv < TILE_WAVES; ++wv) tot += s_dw[tid * TILE_WAVES + wv];
          de.part[(int64_t)bid * de.kd + tid] = tot;
        }
      } else {
  #pragma unroll
      for (int u = 0; u < EPI_UNROLL; ++u) {
        const int r = tid + u * TILE_THREADS;
        if (r < rows_here) {
          double t = c + acc[r];
          if (addend) t += addend[row0 + r];
          if constexpr (FOLD) {
            // t_k = X~ (s.*r_k) + beta t_{k-1}  ( = X~ (s.*p_k): the product is
            // linear and s.*p_k = s.*r_k + beta s.*p_{k-1} ), kept unscaled
            if (fa.k > 0) t = fma(f_beta, tu_pre[u], t);
            fa.tu[row0 + r] = t;
          }
          double v = t;
          if (rowscale) v *= rs_pre[u];
          out[row0 + r] = v;
          tsum += v;
          t2sum += v * t;
        }
      }
      }
      if (out_sum_part) {
        // partial sum of this panel's outputs (feeds the intercept / centring
        // terms of the following Tdot); fixed order: lanes, then waves.
        // twt_off != 0: also the partial of sum_i rowscale_i t_i^2 = <t, Omega t>,
        // the data part of the CG curvature p.Ap (cg_sampler.hip), written
        // twt_off doubles after the sum's slot.
        tsum = wave_allsum(tsum);
        t2sum = wave_allsum(t2sum);
        if constexpr (FOLD) f_pdp = wave_allsum(f_pdp);
        __syncthreads();
        if (lane == 0) {
          xs[wave] = tsum;
          xs[TILE_WAVES + wave] = t2sum;
          if constexpr (FOLD) xs[2 * TILE_WAVES + wave] = f_pdp;
        }
        __syncthreads();
        if (tid == 0) {
          double tot = 0., tot2 = 0., tot3 = 0.;
          for (int wv = 0; wv < TILE_WAVES; ++wv) {
            tot += xs[wv];
            tot2 += xs[TILE_WAVES + wv];
            if constexpr (FOLD) tot3 += xs[2 * TILE_WAVES + wv];
          }
          out_sum_part[bid] = tot;
          if (twt_off) out_sum_part[twt_off + bid] = tot2;
          if constexpr (FOLD) fa.pdp_part[bid] = tot3;   // <p, d p> of this share
        }
        // consumers add NPART slots: the first workgroup clears the unused ones
        if (bid == 0 && (int)gridDim.x + tid < NPART) {
          out_sum_part[gridDim.x + tid] = 0.;
          if (twt_off) out_sum_part[twt_off + (int)gridDim.x + tid] = 0.;
          if constexpr (FOLD) fa.pdp_part[gridDim.x + tid] = 0.;
        }
      }
    } else {
      double* dst = slab + (int64_t)group * R + row0;
      for (int r = tid; r < rows_here; r += TILE_THREADS) dst[r] = acc[r];
    }
  }
  if (dbg && lane == 0) {
    unsigned long long* o = dbg + ((size_t)bid * TILE_WAVES + wave) * 6;
    o[0] = (unsigned)__builtin_amdgcn_s_memtime() - t_start;  // whole wave
    o[1] = t_loop;    // stream loop incl. tile switches
    o[2] = t_switch;  // inside tile switches
    o[3] = t_skew;    // of which: arrival -> every wave arrived
    o[4] = t_drain;   // of which: this wave's own loads (slice values, ring)
  }
}

// out[r] = rowscale[r] * (c + sum_g slab[g][r])   (dot with G > 1)
__global__ __launch_bounds__(256) void tiled_dot_finalize_kernel(
    int64_t R, int G, const double* __restrict__ slab,
    const double* __restrict__ c_part, const double* x0_ptr,
    const double* __restrict__ rowscale, double* __restrict__ out,
    const double* __restrict__ addend) {
  double c = x0_ptr ? *x0_ptr : 0.;
  if (c_part) {
    double cs = 0.;
    for (int k = 0; k < NPART; ++k) cs += c_part[k];
    c -= cs;
  }
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < R;
       r += (int64_t)gridDim.x * 256) {
    double a = 0.;
    for (int g = 0; g < G; ++g) a += slab[(int64_t)g * R + r];
    double v = c + a;
    if (addend) v += addend[r];
    if (rowscale) v *= rowscale[r];
    out[r] = v;
  }
}

// ------------------------------------------------------ upload of a layout

static int upload(DevMem& dst, const void* src, size_t bytes) {
  BBX_TRY(dst.alloc(bytes > 0 ? bytes : 8));
  if (bytes > 0) BBX_HIP(hipMemcpy(dst.ptr, src, bytes, hipMemcpyHostToDevice));
  return BBX_OK;
}

// Builds one orientation on the host (tiled_layout.cpp) and moves it to HBM.
static int build_one(TiledMatrix& m, int64_t R, int64_t C, int64_t nnz,
                     const int64_t* rowptr, const int32_t* colidx,
                     const double* vals, bool transpose, int K) {
  TiledHost host;
  std::string err;
  TiledOptions opt = TiledOptions::from_env(transpose);
  opt.chains = K;
  if (K > 1) opt.force_PR = opt.force_G = 0;  // overrides tune the K = 1 layout
  if (build_tiled_host(R, C, nnz, rowptr, colidx, vals, opt, &host, &err) != 0)
    return fail(BBX_ERR_INVALID, err);
  m.K = host.K;
  m.R = host.R;
  m.C = host.C;
  m.nnz = host.nnz;
  m.W = host.W;
  m.n_block = host.n_block;
  m.PR = host.PR;
  m.n_panel = host.n_panel;
  m.G = host.G;
  m.has_vals = host.has_vals;
  m.packed = host.packed;
  m.Wl = host.Wl;
  m.n_slice = host.n_slice;
  m.n_quad = host.n_quad;
  m.n_tile = host.n_tile;
  m.n_desc = host.n_desc;
  m.desc_stride = host.desc_stride;
  m.n_extra = host.n_extra;
  m.split_T = host.split_T;
  BBX_TRY(upload(m.folds, host.folds.data(), host.folds.size() * sizeof(FoldDesc)));
  BBX_TRY(upload(m.panel_fold, host.panel_fold.data(),
                 host.panel_fold.size() * sizeof(int32_t)));
  BBX_TRY(upload(m.ids, host.ids.data(), host.ids.size() * sizeof(Ids4)));
  if (m.has_vals)
    BBX_TRY(upload(m.vals, host.vals.data(), host.vals.size() * sizeof(double)));
  BBX_TRY(upload(m.descs, host.descs.data(),
                 host.descs.size() * sizeof(BatchDesc)));
  BBX_TRY(upload(m.wave_desc, host.wave_desc.data(),
                 host.wave_desc.size() * sizeof(int32_t)));
  BBX_TRY(upload(m.wg_quad0, host.wg_quad0.data(),
                 host.wg_quad0.size() * sizeof(uint32_t)));
  BBX_TRY(upload(m.rowids, host.rowids.data(),
                 host.rowids.size() * sizeof(uint32_t)));
  BBX_TRY(m.slab.alloc(sizeof(double) * (size_t)m.G * (size_t)R * (size_t)m.K));
  return BBX_OK;
}

static size_t lds_bytes(const TiledMatrix& m) {
  return sizeof(double) * (size_t)m.K *
         ((size_t)m.Wl + 8 + (size_t)m.PR + (size_t)m.n_extra);
}

void destroy_tiled(bbx_design* h) {
  delete static_cast<TiledPair*>(h->tiled);
  h->tiled = nullptr;
  delete static_cast<HybridParts*>(h->hybrid);
  h->hybrid = nullptr;
  for (void*& t : h->tiled_k) {
    delete static_cast<TiledPair*>(t);
    t = nullptr;
  }
}

// A host CSR as the builders read it: HostCsr (tiled_layout.hpp) owns storage,
// CsrRef points into one -- a copy fetched from the device arrays of the
// handle, or the host copy a design of 2^31 or more entries was created from
// (bbx_design::host_csr: such a design never has device CSR arrays).
struct CsrRef {
  const int64_t* rowptr = nullptr;
  const int32_t* colidx = nullptr;
  const double* vals = nullptr;   // nullptr: every stored value is 1.0
};

static int fetch_host_csr(const bbx_design* h, bool transpose, HostCsr* store,
                          CsrRef* ref) {
  if (const HostCsr* held =
          static_cast<const HostCsr*>(h->host_csr[transpose ? 1 : 0])) {
    ref->rowptr = held->rowptr.data();
    ref->colidx = held->colidx.data();
    ref->vals = h->binary ? nullptr : held->vals.data();
    return BBX_OK;
  }
  const int64_t R = transpose ? h->p : h->n, nnz = h->nnz;
  const DevMem& ip = transpose ? h->t_indptr : h->indptr;
  const DevMem& ix = transpose ? h->t_indices : h->indices;
  const DevMem& da = transpose ? h->t_data : h->data;
  if (!ip.ptr || !ix.ptr)
    return fail(BBX_ERR_STATE,
                "the reference-layout arrays of this design are not in HBM (a "
                "design of 2^31 or more stored entries keeps its tiled form "
                "only: one chain at a time, no batch layouts)");
  std::vector<int32_t> narrow((size_t)R + 1);
  BBX_HIP(hipMemcpy(narrow.data(), ip.ptr, sizeof(int32_t) * (size_t)(R + 1),
                    hipMemcpyDeviceToHost));
  store->rowptr.assign(narrow.begin(), narrow.end());
  store->colidx.resize((size_t)std::max<int64_t>(nnz, 1));
  if (nnz > 0)
    BBX_HIP(hipMemcpy(store->colidx.data(), ix.ptr, sizeof(int32_t) * (size_t)nnz,
                      hipMemcpyDeviceToHost));
  store->vals.clear();
  if (!h->binary) {
    store->vals.resize((size_t)std::max<int64_t>(nnz, 1));
    BBX_HIP(hipMemcpy(store->vals.data(), da.ptr, sizeof(double) * (size_t)nnz,
                      hipMemcpyDeviceToHost));
  }
  ref->rowptr = store->rowptr.data();
  ref->colidx = store->colidx.data();
  ref->vals = h->binary ? nullptr : store->vals.data();
  return BBX_OK;
}

static int check_lds(const TiledPair* tp) {
  for (const TiledMatrix* m : {&tp->x, &tp->xt})
    if (lds_bytes(*m) > (size_t)TILE_LDS_BYTES)
      return fail(BBX_ERR_INVALID, "tile does not fit in LDS");
  return BBX_OK;
}

// Builds both orientations from the device CSR arrays in the handle (CSR of X
// and CSR of X^T) through a host pass, sized for K right-hand sides.
static int build_tiled_pair(bbx_design* h, int K, void** slot) {
  const int64_t n = h->n, p = h->p, nnz = h->nnz;
  TiledPair* tp = new (std::nothrow) TiledPair();
  if (!tp) return fail(BBX_ERR_INVALID, "out of host memory");
  *slot = tp;  // owned by the handle from here on (destroy_tiled)
  HostCsr store;
  CsrRef c;
  BBX_TRY(fetch_host_csr(h, false, &store, &c));
  BBX_TRY(build_one(tp->x, n, p, nnz, c.rowptr, c.colidx, c.vals, false, K));
  // transpose orientation from the CSR of X^T (built on the device, or on the
  // host for 64-bit input)
  BBX_TRY(fetch_host_csr(h, true, &store, &c));
  BBX_TRY(build_one(tp->xt, p, n, nnz, c.rowptr, c.colidx, c.vals, true, K));
  return check_lds(tp);
}

// Rows of one orientation split by value (see HybridParts): the entries equal
// to 1.0, and the others outside the dense columns.  `col_is_dense` is indexed
// by the ORIGINAL column id: the row id for the transposed orientation.
static void split_rows(int64_t R, const CsrRef& c, bool transpose,
                       const std::vector<uint8_t>& col_is_dense, HostCsr* ones,
                       HostCsr* rest) {
  ones->rowptr.assign((size_t)R + 1, 0);
  rest->rowptr.assign((size_t)R + 1, 0);
  ones->colidx.clear();
  rest->colidx.clear();
  ones->vals.clear();
  rest->vals.clear();
  for (int64_t r = 0; r < R; ++r) {
    for (int64_t k = c.rowptr[(size_t)r]; k < c.rowptr[(size_t)r + 1]; ++k) {
      const double v = c.vals[(size_t)k];
      const int32_t j = c.colidx[(size_t)k];
      if (v == 1.0) {
        ones->colidx.push_back(j);
      } else if (!col_is_dense[(size_t)(transpose ? r : j)]) {
        rest->colidx.push_back(j);
        rest->vals.push_back(v);
      }
    }
    ones->rowptr[(size_t)r + 1] = (int64_t)ones->colidx.size();
    rest->rowptr[(size_t)r + 1] = (int64_t)rest->colidx.size();
  }
  if (ones->colidx.empty()) ones->colidx.push_back(0);
  if (rest->colidx.empty()) {
    rest->colidx.push_back(0);
    rest->vals.push_back(0.);
  }
}

// Decides whether the design is worth splitting and builds the parts.
// Returns BBX_OK with h->hybrid set, or 1 when the plain valued layout is the
// right one (nothing built).
static int build_hybrid(bbx_design* h) {
  const int64_t n = h->n, p = h->p, nnz = h->nnz;
  if (h->binary || nnz == 0) return 1;
  HostCsr store;
  CsrRef cx;
  BBX_TRY(fetch_host_csr(h, false, &store, &cx));
  // per column: entries equal to 1.0 and other entries
  std::vector<int64_t> c_one((size_t)p, 0), c_val((size_t)p, 0);
  for (int64_t k = 0; k < nnz; ++k)
    (cx.vals[(size_t)k] == 1.0 ? c_one : c_val)[(size_t)cx.colidx[(size_t)k]] += 1;
  // a dense column: at least half of its rows hold a value other than 1.0 (a
  // continuous covariate; it is stored as n doubles, so a sparser column costs
  // more in the block than in the valued layout's 10 bytes per entry)
  const int64_t dense_min = std::max<int64_t>(n / 2, 1);
  std::vector<uint8_t> is_dense((size_t)p, 0);
  std::vector<int32_t> dense_cols;
  int64_t ones_nnz = 0, dense_nnz = 0;
  for (int64_t j = 0; j < p; ++j) {
    ones_nnz += c_one[(size_t)j];
    if (c_val[(size_t)j] >= dense_min) {
      is_dense[(size_t)j] = 1;
      dense_cols.push_back((int32_t)j);
      dense_nnz += c_val[(size_t)j];
    }
  }
  // a dense block beyond 32 GB (it is held twice: by column for the separate
  // products, by row for the operator's single pass) or wider than the
  // single-pass kernels reach (8192 columns: four column pairs per thread)
  // stays in the valued tiled part
  if ((double)dense_cols.size() * (double)n * 8. > 32e9 ||
      dense_cols.size() > (size_t)HYB_FUSED_MAX_KD) {
    std::fill(is_dense.begin(), is_dense.end(), 0);
    dense_cols.clear();
    dense_nnz = 0;
  }
  const int64_t rest_nnz = nnz - ones_nnz - dense_nnz;
  // worth it when the value-free part and the dense block carry most entries
  // (a design without dense columns: when at least a quarter of the entries
  // are ones; with a dense block the block itself is the cheap part -- 1 000
  // Gaussian columns next to 9 000 binary ones, tests/helper.py:13 at scale,
  // are 92 % dense entries and still belong here)
  if (2 * (ones_nnz + dense_nnz) < nnz) return 1;
  // ... and outside the dense block at least a quarter of the entries are ones
  // (what the value-free layout is for; also keeps designs without a single
  // 1.0 in the plain valued layout)
  if (ones_nnz == 0 || 4 * ones_nnz < nnz - dense_nnz) return 1;
  HybridParts* hp = new (std::nothrow) HybridParts();
  if (!hp) return fail(BBX_ERR_INVALID, "out of host memory");
  h->hybrid = hp;  // owned by the handle (destroy_tiled)
  hp->ones_nnz = ones_nnz;
  hp->rest_nnz = rest_nnz;
  hp->dense_nnz = dense_nnz;
  hp->kd = (int)dense_cols.size();
  HostCsr ones, rest;
  split_rows(n, cx, false, is_dense, &ones, &rest);
  BBX_TRY(build_one(hp->ones.x, n, p, ones_nnz, ones.rowptr.data(),
                    ones.colidx.data(), nullptr, false, 1));
  if (rest_nnz > 0)
    BBX_TRY(build_one(hp->rest.x, n, p, rest_nnz, rest.rowptr.data(),
                      rest.colidx.data(), rest.vals.data(), false, 1));
  if (hp->kd > 0) {
    // D column-major: D[slot][i] = the non-one entry of row i in that column
    std::vector<int32_t> slot_of((size_t)p, -1);
    for (int s_ = 0; s_ < hp->kd; ++s_) slot_of[(size_t)dense_cols[(size_t)s_]] = s_;
    std::vector<double> D((size_t)hp->kd * (size_t)n, 0.);
    for (int64_t r = 0; r < n; ++r)
      for (int64_t k = cx.rowptr[(size_t)r]; k < cx.rowptr[(size_t)r + 1]; ++k) {
        const int32_t s_ = slot_of[(size_t)cx.colidx[(size_t)k]];
        // (duplicates of one (row, column) add up, like everywhere else)
        if (s_ >= 0 && cx.vals[(size_t)k] != 1.0)
          D[(size_t)s_ * (size_t)n + (size_t)r] += cx.vals[(size_t)k];
      }
    BBX_TRY(upload(hp->D, D.data(), D.size() * sizeof(double)));
    if (hp->kd > DENSE_EPI_MAX && hp->kd <= HYB_FUSED_MAX_KD) {
      // row-major copy for the single pass of an operator application
      hp->ld_rm = (hp->kd + 1) / 2 * 2;
      std::vector<double> Drm((size_t)n * (size_t)hp->ld_rm, 0.);
      for (int s_ = 0; s_ < hp->kd; ++s_)
        for (int64_t r = 0; r < n; ++r)
          Drm[(size_t)r * (size_t)hp->ld_rm + (size_t)s_] =
              D[(size_t)s_ * (size_t)n + (size_t)r];
      BBX_TRY(upload(hp->D_rm, Drm.data(), Drm.size() * sizeof(double)));
    }
    BBX_TRY(upload(hp->dense_cols, dense_cols.data(),
                   dense_cols.size() * sizeof(int32_t)));
    BBX_TRY(hp->d_part.alloc(sizeof(double) * HYB_TDOT_CHUNKS * (size_t)hp->kd));
  }
  {  // transposed orientation of B and S
    CsrRef ct;
    BBX_TRY(fetch_host_csr(h, true, &store, &ct));
    split_rows(p, ct, true, is_dense, &ones, &rest);
    BBX_TRY(build_one(hp->ones.xt, p, n, ones_nnz, ones.rowptr.data(),
                      ones.colidx.data(), nullptr, true, 1));
    if (rest_nnz > 0)
      BBX_TRY(build_one(hp->rest.xt, p, n, rest_nnz, rest.rowptr.data(),
                        rest.colidx.data(), rest.vals.data(), true, 1));
  }
  BBX_TRY(check_lds(&hp->ones));
  if (rest_nnz > 0) BBX_TRY(check_lds(&hp->rest));
  hp->n_slab = hp->ones.xt.G + (rest_nnz > 0 ? hp->rest.xt.G : 0) +
               (hp->kd > 0 ? 1 : 0);
  BBX_TRY(hp->slab.alloc(sizeof(double) * (size_t)hp->n_slab * (size_t)p));
  BBX_HIP(hipMemset(hp->slab.ptr, 0, sizeof(double) * (size_t)hp->n_slab * (size_t)p));
  BBX_TRY(hp->addend.alloc(sizeof(double) * (size_t)n));
  return BBX_OK;
}

int build_tiled(bbx_design* h) {
  // mixed designs: value-free part + dense block + valued rest (HybridParts)
  const int st_h = no_throw([&]() -> int { return build_hybrid(h); });
  if (st_h < 0) return st_h;
  if (st_h != BBX_OK) BBX_TRY(build_tiled_pair(h, 1, &h->tiled));
#define BBX_TILED_ATTR(VV, WW, KK, FF, PP)                                     \
  BBX_HIP(hipFuncSetAttribute(                                                 \
      reinterpret_cast<const void*>(                                           \
          &tiled_spmv_kernel<VV, WW, KK, FF, false, PP>),                      \
      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
  // (the dense-epilogue instantiations hold 2 KB of static LDS: the layouts
  // leave exactly that free, lds_budget_per_chain)
  BBX_HIP(hipFuncSetAttribute(
      reinterpret_cast<const void*>(
          &tiled_spmv_kernel<false, true, 0, false, true, false>),
      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
  BBX_HIP(hipFuncSetAttribute(
      reinterpret_cast<const void*>(
          &tiled_spmv_kernel<false, true, 0, false, true, true>),
      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
  BBX_TILED_ATTR(false, false, 0, false, false);
  BBX_TILED_ATTR(false, true, 0, false, false);
  BBX_TILED_ATTR(true, false, 0, false, false);
  BBX_TILED_ATTR(true, true, 0, false, false);
  BBX_TILED_ATTR(false, true, 1, false, false);
  BBX_TILED_ATTR(false, true, 2, false, false);
  BBX_TILED_ATTR(true, true, 1, false, false);
  BBX_TILED_ATTR(false, true, 0, true, false);   // folded direction step (CG loop)
  // value-free, one right-hand side, ids packed in groups of five
  BBX_TILED_ATTR(false, false, 0, false, true);
  BBX_TILED_ATTR(false, true, 0, false, true);
  BBX_TILED_ATTR(false, true, 0, true, true);
#undef BBX_TILED_ATTR
  // The reference-layout index arrays stay in HBM (0.8 GB at 1M x 50k, next to
  // 288 GB): a layout sized for K batched chains is built from them on first
  // use (ensure_tiled_k), and storage = 'csr' cross-checks need them anyway.
  return BBX_OK;
}

// K-layout of a mixed design: the value-free part B in its K-layout, the dense
// block D as it is, the valued rest S (if any; pairs only) in its K-layout.
static int build_split_k(bbx_design* h, int K, void** slot) {
  HybridParts* hp = static_cast<HybridParts*>(h->hybrid);
  const int64_t n = h->n, p = h->p;
  const int ks = K == 2 ? 0 : 1;
  TiledPair* tp = new (std::nothrow) TiledPair();
  if (!tp) return fail(BBX_ERR_INVALID, "out of host memory");
  *slot = tp;  // owned by the handle from here on (destroy_tiled)
  std::vector<uint8_t> is_dense((size_t)p, 0);
  if (hp->kd > 0) {
    std::vector<int32_t> cols((size_t)hp->kd);
    BBX_HIP(hipMemcpy(cols.data(), hp->dense_cols.ptr, sizeof(int32_t) * cols.size(),
                      hipMemcpyDeviceToHost));
    for (int32_t j : cols) is_dense[(size_t)j] = 1;
  }
  HostCsr store, ones, rest;
  CsrRef c;
  BBX_TRY(fetch_host_csr(h, false, &store, &c));
  split_rows(n, c, false, is_dense, &ones, &rest);
  BBX_TRY(build_one(tp->x, n, p, hp->ones_nnz, ones.rowptr.data(),
                    ones.colidx.data(), nullptr, false, K));
  if (hp->rest_nnz > 0) {
    if (K != 2) return fail(BBX_ERR_INVALID, "valued designs batch at most 2 chains");
    if (!hp->rest_k) hp->rest_k = new (std::nothrow) TiledPair();
    if (!hp->rest_k) return fail(BBX_ERR_INVALID, "out of host memory");
    BBX_TRY(build_one(hp->rest_k->x, n, p, hp->rest_nnz, rest.rowptr.data(),
                      rest.colidx.data(), rest.vals.data(), false, K));
  }
  BBX_TRY(fetch_host_csr(h, true, &store, &c));
  split_rows(p, c, true, is_dense, &ones, &rest);
  BBX_TRY(build_one(tp->xt, p, n, hp->ones_nnz, ones.rowptr.data(),
                    ones.colidx.data(), nullptr, true, K));
  int G_rest = 0;
  if (hp->rest_nnz > 0) {
    BBX_TRY(build_one(hp->rest_k->xt, p, n, hp->rest_nnz, rest.rowptr.data(),
                      rest.colidx.data(), rest.vals.data(), true, K));
    BBX_TRY(check_lds(hp->rest_k));
    G_rest = hp->rest_k->xt.G;
  }
  BBX_TRY(check_lds(tp));
  const size_t slab_bytes =
      sizeof(double) * (size_t)(tp->xt.G + G_rest + 1) * (size_t)p * (size_t)K;
  BBX_TRY(hp->slab_k[ks].alloc(slab_bytes));
  BBX_HIP(hipMemset(hp->slab_k[ks].ptr, 0, slab_bytes));
  if (hp->addend_k.bytes < sizeof(double) * (size_t)n * 4)
    BBX_TRY(hp->addend_k.alloc(sizeof(double) * (size_t)n * 4));
  if (hp->kd > 0 &&
      hp->d_part_k.bytes < sizeof(double) * HYB_TDOT_CHUNKS * (size_t)hp->kd * 4)
    BBX_TRY(hp->d_part_k.alloc(sizeof(double) * HYB_TDOT_CHUNKS * (size_t)hp->kd * 4));
  hp->split_k[ks] = true;
  return BBX_OK;
}

// Does a batch on this design run through value-free kernels (any width), or
// through the valued ones (pairs only)?
bool tiled_batch_value_free(const bbx_design* h) {
  if (h->binary) return true;
  const HybridParts* hp = static_cast<const HybridParts*>(h->hybrid);
  return hp && hp->rest_nnz == 0;
}

// The cost model's view of a batch of K chains on this design (no layout is
// built): K single-chain products against one K-column product, both
// orientations, from the geometry search's estimates (tiled_layout.cpp
// shape_cost: ~4.3 us per tile + ~24 ps per streamed entry, rounds of 256
// workgroups).  > 1: the batch is predicted to pay.  At 1M x 50k: K = 2 -> 1.08
// (measured 1.47 on the products, 1.35 on whole iterations), K = 4 -> 0.67
// (measured 0.99 / 0.975): the model is pessimistic in absolute terms but
// orders the widths correctly.  Mixed designs are priced on the whole matrix.
int tiled_batch_predict(const bbx_design* h, int K, double* speedup) {
  if (!h->sparse || h->format != BBX_FORMAT_TILED)
    return fail(BBX_ERR_STATE, "batched chains need the tiled format");
  if (K != 2 && K != 4) return fail(BBX_ERR_INVALID, "K must be 2 or 4");
  return no_throw([&]() -> int {
    if (!h->indptr.ptr || !h->t_indptr.ptr)
      return fail(BBX_ERR_STATE,
                  "a design of 2^31 or more stored entries runs one chain at a "
                  "time (its reference-layout arrays are not kept)");
    std::vector<int32_t> xp32((size_t)h->n + 1), tp32((size_t)h->p + 1);
    BBX_HIP(hipMemcpy(xp32.data(), h->indptr.ptr, sizeof(int32_t) * xp32.size(),
                      hipMemcpyDeviceToHost));
    BBX_HIP(hipMemcpy(tp32.data(), h->t_indptr.ptr, sizeof(int32_t) * tp32.size(),
                      hipMemcpyDeviceToHost));
    const std::vector<int64_t> xp(xp32.begin(), xp32.end()),
        tp(tp32.begin(), tp32.end());
    double c1 = 0., ck = 0.;
    for (int k : {1, K}) {
      const double cx = tiled_model_cost(h->n, h->p, h->nnz, xp.data(), k);
      const double ct = tiled_model_cost(h->p, h->n, h->nnz, tp.data(), k);
      if (cx < 0. || ct < 0.) return fail(BBX_ERR_INVALID, "cost model failed");
      (k == 1 ? c1 : ck) = cx + ct;
    }
    *speedup = ck > 0. ? (double)K * c1 / ck : 0.;
    return BBX_OK;
  });
}

// The layout sized for K right-hand sides (K = 2, 4), built on first use.
int ensure_tiled_k(bbx_design* h, int K) {
  // (a mixed design's K-layout is the plain valued one of the whole matrix)
  if (!h->sparse || h->format != BBX_FORMAT_TILED || (!h->tiled && !h->hybrid))
    return fail(BBX_ERR_STATE, "batched chains need the tiled format");
  if (K == 1) return BBX_OK;
  if (K != 2 && K != 4) return fail(BBX_ERR_INVALID, "K must be 1, 2 or 4");
  void** slot = &h->tiled_k[K == 2 ? 0 : 1];
  if (*slot) return BBX_OK;
  HybridParts* hp = static_cast<HybridParts*>(h->hybrid);
  const int st = (hp && (hp->rest_nnz == 0 || K == 2))
                     ? no_throw([&]() -> int { return build_split_k(h, K, slot); })
                     : build_tiled_pair(h, K, slot);
  if (st < 0) {
    delete static_cast<TiledPair*>(*slot);
    *slot = nullptr;
  }
  return st;
}

static const TiledPair* tiled_pair_for(const bbx_design* h, int K) {
  return static_cast<const TiledPair*>(
      K == 1 ? h->tiled : h->tiled_k[K == 2 ? 0 : 1]);
}

static int launch_tiled(bbx_design* h, const TiledMatrix& m, const double* x,
                        const double* c_part, const double* x0_ptr,
                        const double* rowscale, double* out, double* slab,
                        double* out_sum_part, hipEvent_t ev_begin = nullptr,
                        hipEvent_t ev_end = nullptr, int twt_off = 0,
                        const TiledBatchArgs* ba = nullptr,
                        const double* addend = nullptr,
                        const DotFold* fold = nullptr,
                        const DenseEpi* dense = nullptr) {
  const unsigned grid = (unsigned)(m.n_panel * m.G);
  const size_t lb = lds_bytes(m);
  // Instrumented builds only (-DBBX_TILED_INSTRUMENT=1; the product library
  // reads neither variable): BBX_ABLATE=bits removes the gathers / slice loads /
  // switch barriers (timing only, wrong results), BBX_TILED_DEBUG=N prints the
  // per-wave phase timing of launches N and N+1.
#if BBX_TILED_INSTRUMENT
  static const int ablate = getenv("BBX_ABLATE") ? atoi(getenv("BBX_ABLATE")) : 0;
  static const int dbg_at =
      getenv("BBX_TILED_DEBUG") ? atoi(getenv("BBX_TILED_DEBUG")) : -1;
#else
  constexpr int dbg_at = -1;
#endif
  static int dbg_count = 0;
  static unsigned long long* dbg_buf = nullptr;
  unsigned long long* dbg = nullptr;
  if (dbg_at >= 0 && BBX_TILED_INSTRUMENT) {
    if (!dbg_buf)
      BBX_HIP(hipMalloc(&dbg_buf, sizeof(unsigned long long) * 4096 * TILE_WAVES * 6));
    if ((dbg_count == dbg_at || dbg_count == dbg_at + 1) && grid <= 4096)
      dbg = dbg_buf;
    ++dbg_count;
  }
  static const TiledBatchArgs no_batch{};
  const TiledBatchArgs& bb = ba ? *ba : no_batch;
  const DotFold fa = fold ? *fold : DotFold{};
  const DenseEpi de = dense ? *dense : DenseEpi{};
#define BBX_TILED_LAUNCH_W(VV, WW, KK, VALPTR)                                 \
  BBX_TILED_LAUNCH_D(VV, WW, KK, false, false, false, VALPTR)
#define BBX_TILED_LAUNCH_D(VV, WW, KK, FF, DD, PP, VALPTR)                     \
  BBX_LAUNCH_EXT((tiled_spmv_kernel<VV, WW, KK, FF, DD, PP>),           \
                     dim3(grid),                                               \
                     dim3(TILE_THREADS), (unsigned)lb, h->stream, ev_begin,    \
                     ev_end, 0u, m.R, m.C, m.W, m.PR,                          \
                     m.G, (m.n_block + m.G - 1) / m.G,                         \
                     m.wave_desc.as<int32_t>(), m.desc_stride,                 \
                     m.descs.as<BatchDesc>(), m.rowids.as<uint32_t>(),         \
                     m.wg_quad0.as<uint32_t>(),                                \
                     m.ids.as<uint4>(), VALPTR, x, c_part, x0_ptr, rowscale,   \
                     out, slab, m.PR + m.n_extra,                              \
                     m.panel_fold.as<int32_t>(), m.folds.as<FoldDesc>(),       \
                     out_sum_part, twt_off, BBX_INSTR_ARGS h->skip_flag,       \
                     bb.rowscale, bb.out, bb.out_stride, bb.part_stride,       \
                     addend, fa, de)
// value-free, one right-hand side: plain ids or packed groups (m.packed)
#define BBX_TILED_LAUNCH_P(WW, FF, DD)                                         \
  do {                                                                         \
    if (m.packed) BBX_TILED_LAUNCH_D(false, WW, 0, FF, DD, true, nullptr);     \
    else BBX_TILED_LAUNCH_D(false, WW, 0, FF, DD, false, nullptr);             \
  } while (0)
  // 16-byte slice loads need every slice start 16-byte aligned: W is a multiple
  // of 64 doubles, so it is the alignment of x itself that decides (inside the
  // CG loop x is an internal buffer placed accordingly; a caller's v + 1 of a
  // design with intercept is not, and takes the 8-byte path)
  const bool wide = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  if (m.K > 1 && !wide)
    return fail(BBX_ERR_INVALID, "batched input must be 16-byte aligned");
  // (four valued right-hand sides do not fit the 128-VGPR budget of a
  // 1024-thread workgroup without scratch: valued designs batch two chains)
  if (m.K == 4 && m.has_vals)
    return fail(BBX_ERR_INVALID, "valued designs batch at most 2 chains");
  if (dense) {
    // mixed design inside an operator application: the dense block in the
    // value-free kernel's epilogue (single chain, direct epilogue)
    // (the epilogue writes one row of kd partials per workgroup into a buffer
    // of NPART rows: the launch site guards the grid, not only its caller)
    if (m.K != 1 || !wide || m.G != 1 || !out || m.has_vals || addend || fold ||
        dense->kd < 1 || dense->kd > DENSE_EPI_MAX || grid > (unsigned)NPART)
      return fail(BBX_ERR_STATE, "dense epilogue: unsupported launch");
    BBX_TILED_LAUNCH_P(true, false, true);
  } else if (fold) {
    // folded direction step: single chain, 16-byte aligned vectors, the direct
    // epilogue with its partial sums
    if (m.K != 1 || !wide || m.G != 1 || !out || !out_sum_part || addend)
      return fail(BBX_ERR_STATE, "folded direction step: unsupported launch");
    // (with stored values the two fill vectors do not fit next to the value
    // ring: 16 VGPRs would spill to scratch under an asm-issued ring)
    if (m.has_vals)
      return fail(BBX_ERR_STATE, "folded direction step: value-free layout only");
    BBX_TILED_LAUNCH_P(true, true, false);
  } else if (m.K == 4)
    BBX_TILED_LAUNCH_W(false, true, 2, nullptr);
  else if (m.has_vals) {
    if (m.K == 2) BBX_TILED_LAUNCH_W(true, true, 1, m.vals.as<double>());
    else if (wide) BBX_TILED_LAUNCH_W(true, true, 0, m.vals.as<double>());
    else BBX_TILED_LAUNCH_W(true, false, 0, m.vals.as<double>());
  } else if (m.K == 2)
    BBX_TILED_LAUNCH_W(false, true, 1, nullptr);
  else if (wide)
    BBX_TILED_LAUNCH_P(true, false, false);
  else
    BBX_TILED_LAUNCH_P(false, false, false);
#undef BBX_TILED_LAUNCH_W
#undef BBX_TILED_LAUNCH_P
#undef BBX_TILED_LAUNCH_D
  BBX_HIP(hipGetLastError());
  if (dbg) {
    BBX_HIP(hipStreamSynchronize(h->stream));
    const size_t nw = (size_t)grid * TILE_WAVES;
    std::vector<unsigned long long> hb(nw * 6);
    BBX_HIP(hipMemcpy(hb.data(), dbg, hb.size() * 8, hipMemcpyDeviceToHost));
    double tot = 0., loop = 0., sw = 0., skew = 0., drain = 0.;
    for (size_t w = 0; w < nw; ++w) {
      tot += (double)hb[w * 6];
      loop += (double)hb[w * 6 + 1];
      sw += (double)hb[w * 6 + 2];
      skew += (double)hb[w * 6 + 3];
      drain += (double)hb[w * 6 + 4];
    }
    fprintf(stderr,
            "[bbx tiled dbg grid=%u launch=%d] ticks per wave (mean): total "
            "%.0f = streaming %.0f + tile switches %.0f (waiting for the last "
            "wave %.0f -- of which the wave's own slice and ring loads %.0f --, "
            "refill + second barrier %.0f) + epilogue %.0f\n",
            grid, dbg_count, tot / nw, (loop - sw) / nw, sw / nw, skew / nw,
            drain / nw, (sw - skew) / nw, (tot - loop) / nw);
  }
  return BBX_OK;
}

// ---- mixed designs (HybridParts): the small kernels around the tiled ones

// addend[i] = sum_g rest_slab[g][i] + sum_j D[j][i] v[intercept + dense_cols[j]]
// (fixed order)
__global__ __launch_bounds__(256) void hyb_addend_kernel(
    int64_t n, int kd, int intercept, const double* __restrict__ D,
    const int32_t* __restrict__ dense_cols, const double* __restrict__ v,
    const double* __restrict__ rest_slab, int G_rest,
    double* __restrict__ addend, const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  extern __shared__ double s_v[];
  for (int j = threadIdx.x; j < kd; j += 256) s_v[j] = v[intercept + dense_cols[j]];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256) {
    double a = 0.;
    for (int g = 0; g < G_rest; ++g) a += rest_slab[(int64_t)g * n + i];
    int j = 0;
    for (; j + 4 <= kd; j += 4) {
      const double d0 = D[(int64_t)j * n + i], d1 = D[(int64_t)(j + 1) * n + i],
                   d2 = D[(int64_t)(j + 2) * n + i], d3 = D[(int64_t)(j + 3) * n + i];
      a += d0 * s_v[j];
      a += d1 * s_v[j + 1];
      a += d2 * s_v[j + 2];
      a += d3 * s_v[j + 3];
    }
    for (; j < kd; ++j) a += D[(int64_t)j * n + i] * s_v[j];
    addend[i] = a;
  }
}

// part[c][j] = sum over row chunk c of D[j][i] w[i]: one WAVE per (column,
// chunk), contiguous 16-byte lane loads, no workgroup synchronisation
__global__ __launch_bounds__(256) void hyb_dense_tdot_kernel(
    int64_t n, int kd, int n_chunk, const double* __restrict__ D,
    const double* __restrict__ w, double* __restrict__ part,
    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t task = (int64_t)blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  if (task >= (int64_t)kd * n_chunk) return;
  const int j = (int)(task / n_chunk), c = (int)(task - (int64_t)j * n_chunk);
  int64_t rows = (n + n_chunk - 1) / n_chunk;
  rows = (rows + 1) / 2 * 2;  // even: chunks start 16-byte aligned
  const int64_t r0 = (int64_t)c * rows;
  const int64_t r1 = (r0 + rows < n) ? r0 + rows : n;
  const double* __restrict__ Dj = D + (int64_t)j * n;
  double a0 = 0., a1 = 0.;
  int64_t i = r0 + 2 * lane;
  const bool aligned = (n & 1) == 0;  // column starts stay 16-byte aligned
  if (aligned) {
    for (; i + 1 < r1; i += 2 * WAVE) {
      const v2d d = *reinterpret_cast<const v2d*>(Dj + i);
      const v2d ww = *reinterpret_cast<const v2d*>(w + i);
      a0 += d.x * ww.x;
      a1 += d.y * ww.y;
    }
    if (i < r1) a0 += Dj[i] * w[i];
  } else {
    for (i = r0 + lane; i < r1; i += WAVE) a0 += Dj[i] * w[i];
  }
  const double t = wave_allsum(a0 + a1);
  if (lane == 0) part[(int64_t)c * kd + j] = t;
}

// slab_row[dense_cols[j]] = sum_c part[c][j]: one wave per dense column, lanes
// over the chunks, fixed order (a serial sum of 256 dependent loads took 33 us)
__global__ __launch_bounds__(WAVE) void hyb_dense_scatter_kernel(
    int kd, int n_chunk, const int32_t* __restrict__ dense_cols,
    const double* __restrict__ part, double* __restrict__ slab_row,
    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  const int j = blockIdx.x;
  double a = 0.;
  for (int c = threadIdx.x; c < n_chunk; c += WAVE) a += part[(int64_t)c * kd + j];
  a = wave_allsum(a);
  if (threadIdx.x == 0) slab_row[dense_cols[j]] = a;
}

// ---- the dense block of a mixed design inside an OPERATOR APPLICATION, in one
// pass over D (kd > DENSE_EPI_MAX; up to 8 columns ride in the value-free
// kernel's epilogue, DenseEpi):
//   tt_i = a_i + sum_j D[i][j] v_j,   t_i = Omega_i tt_i,   dw_j += D[i][j] t_i
// with a = c + B v from the value-free kernel.  The reference's test designs
// (tests/helper.py:13: binary_frac = .9; simulate_data.py:29-63) carry hundreds
// of continuous columns: with the two separate kernels (hyb_addend_kernel,
// hyb_dense_tdot_kernel) D is streamed twice per application.
// One WAVE per row: lane l owns the column pairs q = l + 64 k, k < GW, of the
// row-major copy -- every load instruction of a wave reads 1 KiB of contiguous
// bytes -- with its slices of v and of D^T t in registers; a row costs GW loads,
// one wave sum and GW rank-1 updates, RB rows in flight, no workgroup
// synchronisation inside the stream.  256 workgroups of 16 waves own
// contiguous row ranges; the per-wave D^T t are added through LDS in wave order
// (fixed order: bitwise reproducible), one row of kd per workgroup for
// hyb_dense_scatter_kernel, and the workgroup's sum t and <t, tt> go where the
// value-free kernel's epilogue would have put them.
typedef double hyb_d2 __attribute__((ext_vector_type(2)));
// (GW = 8, 513-1024 columns: 64 doubles of a lane's registers hold v, D^T t and
// two rows, so the workgroup is 512 threads -- 256 VGPRs per lane)
template <int GW, int RB, int NT>
__global__ __launch_bounds__(NT) void hyb_dense_fused_kernel(
    int64_t n, int kd, int ld_rm, const double* __restrict__ Drm,
    const int32_t* __restrict__ cols, int intercept,
    const double* __restrict__ v, const double* __restrict__ a,
    const double* __restrict__ rowscale, double* __restrict__ t_out,
    double* __restrict__ sum_part, int twt_off, double* __restrict__ dw_part,
    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  extern __shared__ double s_dw[];   // [NW][2 * 64 * GW] + 2 * NW
  constexpr int NW = NT / WAVE;
  constexpr int WCOLS = 2 * WAVE * GW;     // columns a wave's registers cover
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int ldp = ld_rm / 2;
  hyb_d2 vo[GW], g[GW];
  bool has[GW];
#pragma unroll
  for (int k = 0; k < GW; ++k) {
    const int q = lane + WAVE * k;
    has[k] = q < ldp;
    vo[k] = hyb_d2{0., 0.};
    g[k] = hyb_d2{0., 0.};
    if (has[k]) {
      if (2 * q < kd) vo[k].x = v[intercept + cols[2 * q]];
      if (2 * q + 1 < kd) vo[k].y = v[intercept + cols[2 * q + 1]];
    }
  }
  // contiguous rows per workgroup, then per wave
  const int64_t per_wg = (n + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = (int64_t)blockIdx.x * per_wg;
  const int64_t b1 = (b0 + per_wg < n) ? b0 + per_wg : n;
  const int64_t per_wave = (per_wg + NW - 1) / NW;
  const int64_t r0 = b0 + (int64_t)wave * per_wave;
  const int64_t r1 = (r0 + per_wave < b1) ? r0 + per_wave : b1;
  const hyb_d2* __restrict__ D2 = reinterpret_cast<const hyb_d2*>(Drm);
  hyb_d2 xc[RB][GW], xn[RB][GW];
  auto load_rows = [&](int64_t r, hyb_d2 (&x)[RB][GW]) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int k = 0; k < GW; ++k)
        x[i][k] = (r + i < r1 && has[k])
                      ? __builtin_nontemporal_load(D2 + (r + i) * ldp + lane + WAVE * k)
                      : hyb_d2{0., 0.};
  };
  double tsum = 0., t2sum = 0.;
  if (r0 < r1) load_rows(r0, xc);
  for (int64_t r = r0; r < r1; r += RB) {
    load_rows(r + RB, xn);   // in flight across the sums below
    double tt[RB], tv[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      double d0 = 0., d1 = 0.;
#pragma unroll
      for (int k = 0; k < GW; ++k) {
        d0 = fma(xc[i][k].x, vo[k].x, d0);
        d1 = fma(xc[i][k].y, vo[k].y, d1);
      }
      tt[i] = wave_allsum(d0 + d1);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const bool ok = r + i < r1;
      tt[i] = ok ? a[r + i] + tt[i] : 0.;
      tv[i] = (ok && rowscale) ? rowscale[r + i] * tt[i] : tt[i];
      if (ok && lane == i) t_out[r + i] = tv[i];
      tsum += tv[i];
      t2sum = fma(tv[i], tt[i], t2sum);
#pragma unroll
      for (int k = 0; k < GW; ++k) {
        g[k].x = fma(xc[i][k].x, tv[i], g[k].x);
        g[k].y = fma(xc[i][k].y, tv[i], g[k].y);
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int k = 0; k < GW; ++k) xc[i][k] = xn[i][k];
  }
  // the workgroup's D^T t, sum t and <t, tt>: waves in order
#pragma unroll
  for (int k = 0; k < GW; ++k) {
    const int q = lane + WAVE * k;
    s_dw[wave * WCOLS + 2 * q] = g[k].x;
    s_dw[wave * WCOLS + 2 * q + 1] = g[k].y;
  }
  double* s_sc = s_dw + NW * WCOLS;
  if (lane == 0) {
    s_sc[wave] = tsum;
    s_sc[NW + wave] = t2sum;
  }
  __syncthreads();
  for (int j = tid; j < kd; j += NT) {
    double tot = 0.;
    for (int w = 0; w < NW; ++w) tot += s_dw[w * WCOLS + j];
    dw_part[(int64_t)blockIdx.x * kd + j] = tot;
  }
  if (tid == 0 && sum_part) {
    double tot = 0., tot2 = 0.;
    for (int w = 0; w < NW; ++w) {
      tot += s_sc[w];
      tot2 += s_sc[NW + w];
    }
    sum_part[blockIdx.x] = tot;
    if (twt_off) sum_part[twt_off + blockIdx.x] = tot2;
  }
}

// The same pass for WIDER dense blocks (1024 < kd <= 8192 columns): a row no
// longer fits the registers of one wave, so a 1024-thread workgroup takes RB rows
// at a time (4 up to 2048 columns, 2 up to 4096, 1 beyond: 32-64 KB per barrier) -- thread t owns the column pairs t + 1024 k, k < G (1 KiB of
// contiguous bytes per wave and load instruction) with its slices of v_D and of
// D^T t in registers, the rows' inner products go through one LDS exchange and
// ONE barrier per RB rows, the next RB are in flight meanwhile -- the shape of
// the dense designs' single-pass operator (dense.hip dense_fused_f64_kernel).
// Every thread owns its columns outright: the workgroup's row of D^T t partials
// is written without a reduction.
template <int G, int RB>
__global__ __launch_bounds__(1024) void hyb_dense_fused_wg_kernel(
    int64_t n, int kd, int ld_rm, const double* __restrict__ Drm,
    const int32_t* __restrict__ cols, int intercept,
    const double* __restrict__ v, const double* __restrict__ a,
    const double* __restrict__ rowscale, double* __restrict__ t_out,
    double* __restrict__ sum_part, int twt_off, double* __restrict__ dw_part,
    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  constexpr int NW = 1024 / WAVE;
  __shared__ double red[2][RB][NW];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int ldp = ld_rm / 2;
  hyb_d2 vo[G], g[G];
  bool has[G];
#pragma unroll
  for (int k = 0; k < G; ++k) {
    const int q = tid + 1024 * k;
    has[k] = q < ldp;
    vo[k] = hyb_d2{0., 0.};
    g[k] = hyb_d2{0., 0.};
    if (has[k]) {
      if (2 * q < kd) vo[k].x = v[intercept + cols[2 * q]];
      if (2 * q + 1 < kd) vo[k].y = v[intercept + cols[2 * q + 1]];
    }
  }
  const int64_t per_wg = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per_wg;
  const int64_t r1 = (r0 + per_wg < n) ? r0 + per_wg : n;
  const hyb_d2* __restrict__ D2 = reinterpret_cast<const hyb_d2*>(Drm);
  hyb_d2 xc[RB][G], xn[RB][G];
  double sc[RB], sn[RB], ac[RB], an[RB];
  auto load_rows = [&](int64_t r, hyb_d2 (&x)[RB][G], double (&sr)[RB],
                       double (&ar)[RB]) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const bool ok = r + i < r1;
      sr[i] = ok ? (rowscale ? rowscale[r + i] : 1.) : 0.;
      ar[i] = ok ? a[r + i] : 0.;
#pragma unroll
      for (int k = 0; k < G; ++k)
        x[i][k] = (ok && has[k])
                      ? __builtin_nontemporal_load(D2 + (r + i) * ldp + tid + 1024 * k)
                      : hyb_d2{0., 0.};
    }
  };
  double tsum = 0., t2sum = 0.;
  if (r0 < r1) load_rows(r0, xc, sc, ac);
  int buf = 0;
  for (int64_t r = r0; r < r1; r += RB) {
    load_rows(r + RB, xn, sn, an);   // in flight across the exchange below
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      double d0 = 0., d1 = 0.;
#pragma unroll
      for (int k = 0; k < G; ++k) {
        d0 = fma(xc[i][k].x, vo[k].x, d0);
        d1 = fma(xc[i][k].y, vo[k].y, d1);
      }
      const double w = wave_allsum(d0 + d1);
      if (lane == 0) red[buf][i][wave] = w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      double dot = 0.;
#pragma unroll
      for (int w = 0; w < NW; ++w) dot += red[buf][i][w];   // fixed order
      const bool ok = r + i < r1;
      const double tt = ok ? ac[i] + dot : 0.;
      const double tv = sc[i] * tt;      // (sc = 0 past the range)
      if (ok && tid == i) t_out[r + i] = tv;
      tsum += tv;
      t2sum = fma(tv, tt, t2sum);
#pragma unroll
      for (int k = 0; k < G; ++k) {
        g[k].x = fma(xc[i][k].x, tv, g[k].x);
        g[k].y = fma(xc[i][k].y, tv, g[k].y);
      }
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
#pragma unroll
  for (int k = 0; k < G; ++k) {
    const int q = tid + 1024 * k;
    if (2 * q < kd) dw_part[(int64_t)blockIdx.x * kd + 2 * q] = g[k].x;
    if (2 * q + 1 < kd) dw_part[(int64_t)blockIdx.x * kd + 2 * q + 1] = g[k].y;
  }
  if (tid == 0 && sum_part) {   // (every thread holds the same sums)
    sum_part[blockIdx.x] = tsum;
    if (twt_off) sum_part[twt_off + blockIdx.x] = t2sum;
  }
}

// ---- the same for K interleaved right-hand sides: addend[i][c] =
// sum_g rest_slab[g][i][c] + sum_j D[j][i] v[intercept + dense_cols[j]][c]
template <int K>
__global__ __launch_bounds__(256) void hyb_addend_k_kernel(
    int64_t n, int kd, int intercept, const double* __restrict__ D,
    const int32_t* __restrict__ dense_cols, const double* __restrict__ v_il,
    const double* __restrict__ rest_slab, int G_rest,
    double* __restrict__ addend, const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  extern __shared__ double s_v[];   // [kd][K]
  for (int t = threadIdx.x; t < kd * K; t += 256)
    s_v[t] = v_il[(int64_t)(intercept + dense_cols[t / K]) * K + (t % K)];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256) {
    double a[K];
#pragma unroll
    for (int c = 0; c < K; ++c) a[c] = 0.;
    for (int g = 0; g < G_rest; ++g) {
#pragma unroll
      for (int c = 0; c < K; ++c) a[c] += rest_slab[((int64_t)g * n + i) * K + c];
    }
    for (int j = 0; j < kd; ++j) {
      const double d = D[(int64_t)j * n + i];
#pragma unroll
      for (int c = 0; c < K; ++c) a[c] += d * s_v[j * K + c];
    }
#pragma unroll
    for (int c = 0; c < K; ++c) addend[i * K + c] = a[c];
  }
}

// part[(chunk kd + j) K + c] = sum over the chunk's rows of D[j][i] w[i][c]:
// one wave per (column, chunk)
template <int K>
__global__ __launch_bounds__(256) void hyb_dense_tdot_k_kernel(
    int64_t n, int kd, int n_chunk, const double* __restrict__ D,
    const double* __restrict__ w_il, double* __restrict__ part,
    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t task = (int64_t)blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  if (task >= (int64_t)kd * n_chunk) return;
  const int j = (int)(task / n_chunk), ch = (int)(task - (int64_t)j * n_chunk);
  const int64_t rows = (n + n_chunk - 1) / n_chunk;
  const int64_t r0 = (int64_t)ch * rows;
  const int64_t r1 = (r0 + rows < n) ? r0 + rows : n;
  const double* __restrict__ Dj = D + (int64_t)j * n;
  double a[K];
#pragma unroll
  for (int c = 0; c < K; ++c) a[c] = 0.;
  for (int64_t i = r0 + lane; i < r1; i += WAVE) {
    const double d = Dj[i];
#pragma unroll
    for (int c = 0; c < K; ++c) a[c] += d * w_il[i * K + c];
  }
#pragma unroll
  for (int c = 0; c < K; ++c) {
    const double t = wave_allsum(a[c]);
    if (lane == 0) part[((int64_t)ch * kd + j) * K + c] = t;
  }
}

// slab_row[dense_cols[j]][c] = sum over the chunks, fixed order
template <int K>
__global__ __launch_bounds__(WAVE) void hyb_dense_scatter_k_kernel(
    int kd, int n_chunk, const int32_t* __restrict__ dense_cols,
    const double* __restrict__ part, double* __restrict__ slab_row,
    const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag) return;
  const int j = blockIdx.x;
#pragma unroll
  for (int c = 0; c < K; ++c) {
    double a = 0.;
    for (int ch = threadIdx.x; ch < n_chunk; ch += WAVE)
      a += part[((int64_t)ch * kd + j) * K + c];
    a = wave_allsum(a);
    if (threadIdx.x == 0) slab_row[(int64_t)dense_cols[j] * K + c] = a;
  }
}

static int launch_dot_hybrid(bbx_design* h, const double* d_v,
                             const double* d_rowscale, double* d_t,
                             double* d_sum_part, int* sum_done,
                             double* d_twt_part, int* twt_done) {
  HybridParts* hp = static_cast<HybridParts*>(h->hybrid);
  const double* x = d_v + h->intercept;
  const double* x0 = h->intercept ? d_v : nullptr;
  hp->dw_for = nullptr;
  // Inside an operator application the dense block rides in the value-free
  // kernel's epilogue (DenseEpi): no addend kernel, and the partials of
  // D^T (Omega t) for the transposed product that follows.
  {
    const TiledMatrix& mb = hp->ones.x;
    const bool wide = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
    // (A/B: BBX_DENSE_EPI_MAX=0 keeps the separate kernels)
    static const int epi_max = getenv("BBX_DENSE_EPI_MAX")
        ? std::min(atoi(getenv("BBX_DENSE_EPI_MAX")), DENSE_EPI_MAX) : DENSE_EPI_MAX;
    if (h->in_operator && d_rowscale && hp->rest_nnz == 0 && hp->kd > 0 &&
        hp->kd <= epi_max && mb.G == 1 && mb.n_panel <= NPART && wide &&
        d_sum_part) {
      if (!hp->dw_part.ptr)
        BBX_TRY(hp->dw_part.alloc(sizeof(double) * NPART * (size_t)hp->kd));
      DenseEpi de;
      de.D = hp->D.as<double>();
      de.cols = hp->dense_cols.as<int32_t>();
      de.kd = hp->kd;
      de.n = h->n;
      de.part = hp->dw_part.as<double>();
      int twt_off = 0;
      if (sum_done) *sum_done = 1;
      if (d_twt_part && twt_done && d_twt_part != d_sum_part) {
        twt_off = (int)(d_twt_part - d_sum_part);
        *twt_done = 1;
      }
      hipEvent_t ea, eb;
      BBX_TRY(timer_arm(h, 0, &ea, &eb));
      BBX_TRY(launch_tiled(h, mb, x, part_slot(h, PS_C), x0, d_rowscale, d_t,
                           nullptr, d_sum_part, ea, eb, twt_off, nullptr,
                           nullptr, nullptr, &de));
      hp->dw_for = d_t;
      hp->dw_serial = h->operator_serial;
      hp->dw_chunks = mb.n_panel * mb.G;
      return BBX_OK;
    }
  }
  {
    // wider dense blocks inside an operator application: a = c + B v from the
    // value-free kernel, then ONE pass over the row-major copy of D for
    // t = Omega (a + D v_D), its partial sums and D^T t
    const TiledMatrix& mb = hp->ones.x;
    const bool wide = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
    static const bool fused_on = !(getenv("BBX_HYB_FUSED") &&
                                   atoi(getenv("BBX_HYB_FUSED")) == 0);
    if (fused_on && h->in_operator && d_rowscale && hp->rest_nnz == 0 &&
        hp->D_rm.ptr && mb.G == 1 && mb.n_panel <= NPART && wide && d_sum_part) {
      if (!hp->dw_part.ptr)
        BBX_TRY(hp->dw_part.alloc(sizeof(double) * NPART * (size_t)hp->kd));
      int twt_off = 0;
      if (sum_done) *sum_done = 1;
      if (d_twt_part && twt_done && d_twt_part != d_sum_part) {
        twt_off = (int)(d_twt_part - d_sum_part);
        *twt_done = 1;
      }
      BBX_TRY(timer_begin(h, 0));
      BBX_TRY(launch_tiled(h, mb, x, part_slot(h, PS_C), x0, nullptr,
                           hp->addend.as<double>(), nullptr, nullptr));
      const int ldp = hp->ld_rm / 2;
      if (hp->kd > HYB_FUSED_WAVE_KD) {
        // a row of more than 1024 columns: the workgroup-per-row-pair kernel
#define BBX_HYB_WG(GG, RR)                                                     \
  BBX_LAUNCH((hyb_dense_fused_wg_kernel<GG, RR>), dim3(NPART),         \
                     dim3(1024),                                               \
                     0, h->stream, h->n, hp->kd, hp->ld_rm,                    \
                     hp->D_rm.as<double>(), hp->dense_cols.as<int32_t>(),      \
                     h->intercept, d_v, hp->addend.as<double>(), d_rowscale,   \
                     d_t, d_sum_part, twt_off, hp->dw_part.as<double>(),       \
                     h->skip_flag)
        if (ldp <= 1024) BBX_HYB_WG(1, 4);
        else if (ldp <= 2048) BBX_HYB_WG(2, 2);
        else BBX_HYB_WG(4, 1);
#undef BBX_HYB_WG
        BBX_HIP(hipGetLastError());
        hp->dw_for = d_t;
        hp->dw_serial = h->operator_serial;
        hp->dw_chunks = NPART;
        return timer_end(h, 0);
      }
      const int gw = ldp <= WAVE ? 1 : ldp <= 2 * WAVE ? 2 : ldp <= 4 * WAVE ? 4 : 8;
      const int nt = gw == 8 ? 512 : 1024;
      // (the widest forms hold 64 KB + of per-wave partials in dynamic LDS)
      if (!hp->fused_attr_set) {
        BBX_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&hyb_dense_fused_kernel<4, 2, 1024>),
            hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        BBX_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&hyb_dense_fused_kernel<8, 1, 512>),
            hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        hp->fused_attr_set = true;
      }
      const size_t lds =
          sizeof(double) * (size_t)((nt / WAVE) * 2 * WAVE * gw + 2 * (nt / WAVE));
#define BBX_HYB_FUSED(GG, RR, NN)                                              \
  BBX_LAUNCH((hyb_dense_fused_kernel<GG, RR, NN>), dim3(NPART),        \
                     dim3(NN), lds, h->stream, h->n, hp->kd, hp->ld_rm,        \
                     hp->D_rm.as<double>(), hp->dense_cols.as<int32_t>(),      \
                     h->intercept, d_v, hp->addend.as<double>(), d_rowscale,   \
                     d_t, d_sum_part, twt_off, hp->dw_part.as<double>(),       \
                     h->skip_flag)
      if (gw == 1) BBX_HYB_FUSED(1, 4, 1024);
      else if (gw == 2) BBX_HYB_FUSED(2, 4, 1024);
      else if (gw == 4) BBX_HYB_FUSED(4, 2, 1024);
      else BBX_HYB_FUSED(8, 1, 512);
#undef BBX_HYB_FUSED
      BBX_HIP(hipGetLastError());
      hp->dw_for = d_t;
      hp->dw_serial = h->operator_serial;
      hp->dw_chunks = NPART;
      return timer_end(h, 0);
    }
  }
  BBX_TRY(timer_begin(h, 0));
  const double* rest_slab = nullptr;
  int G_rest = 0;
  if (hp->rest_nnz > 0) {
    const TiledMatrix& ms = hp->rest.x;
    BBX_TRY(launch_tiled(h, ms, x, nullptr, nullptr, nullptr, nullptr,
                         ms.slab.as<double>(), nullptr));
    rest_slab = ms.slab.as<double>();
    G_rest = ms.G;
  }
  BBX_LAUNCH(hyb_addend_kernel, dim3(1024), dim3(256),
                     sizeof(double) * (size_t)(hp->kd + 1), h->stream, h->n,
                     hp->kd, h->intercept, hp->D.as<double>(),
                     hp->dense_cols.as<int32_t>(), d_v, rest_slab, G_rest,
                     hp->addend.as<double>(), h->skip_flag);
  BBX_HIP(hipGetLastError());
  const TiledMatrix& mb = hp->ones.x;
  if (mb.G > 1 || mb.n_panel > NPART) {
    // several column groups: slabs, then the finalize kernel adds the addend
    BBX_TRY(launch_tiled(h, mb, x, nullptr, nullptr, nullptr, nullptr,
                         mb.slab.as<double>(), nullptr));
    BBX_LAUNCH(tiled_dot_finalize_kernel, dim3(1024), dim3(256), 0,
                       h->stream, mb.R, mb.G, mb.slab.as<double>(),
                       part_slot(h, PS_C), x0, d_rowscale, d_t,
                       hp->addend.as<double>());
    BBX_HIP(hipGetLastError());
    return timer_end(h, 0);
  }
  double* fused = nullptr;
  int twt_off = 0;
  if (d_sum_part) {
    fused = d_sum_part;
    if (sum_done) *sum_done = 1;
    if (d_twt_part && twt_done && d_twt_part != d_sum_part) {
      twt_off = (int)(d_twt_part - d_sum_part);
      *twt_done = 1;
    }
  }
  BBX_TRY(launch_tiled(h, mb, x, part_slot(h, PS_C), x0, d_rowscale, d_t,
                       nullptr, fused, nullptr, nullptr, twt_off, nullptr,
                       hp->addend.as<double>()));
  return timer_end(h, 0);
}

static int launch_tdot_hybrid(bbx_design* h, const double* d_w,
                              const double* d_sumw_part,
                              const TdotEpilogue& ep, double* d_out) {
  HybridParts* hp = static_cast<HybridParts*>(h->hybrid);
  double* slab = hp->slab.as<double>();
  BBX_TRY(timer_begin(h, 1));
  // (The dense block's kernels on a second stream, beside the tiled kernel,
  // were measured: 65 -> 80 us with five dense columns, 102 -> 175 us with
  // twenty -- they take CUs the one-workgroup-per-CU tiled kernel then waits
  // for.  So: one stream, the tiled kernels first.)
  const TiledMatrix& mb = hp->ones.xt;
  BBX_TRY(launch_tiled(h, mb, d_w, nullptr, nullptr, nullptr, nullptr, slab,
                       nullptr));
  int at = mb.G;
  if (hp->rest_nnz > 0) {
    const TiledMatrix& ms = hp->rest.xt;
    BBX_TRY(launch_tiled(h, ms, d_w, nullptr, nullptr, nullptr, nullptr,
                         slab + (size_t)at * (size_t)h->p, nullptr));
    at += ms.G;
  }
  // (valid for the Tdot of the SAME operator application only; a kernel that
  // rewrote t between the two products would have to clear dw_for)
  if (hp->kd > 0 && hp->dw_for == d_w && h->in_operator &&
      hp->dw_serial == h->operator_serial) {
    // the preceding X~ v kernel of this operator application left the partials
    // of D^T w, one row per workgroup: only the fixed-order sum is left to do
    BBX_LAUNCH(hyb_dense_scatter_kernel, dim3((unsigned)hp->kd),
                       dim3(WAVE), 0, h->stream, hp->kd, hp->dw_chunks,
                       hp->dense_cols.as<int32_t>(), hp->dw_part.as<double>(),
                       slab + (size_t)at * (size_t)h->p, h->skip_flag);
    BBX_HIP(hipGetLastError());
    at += 1;
  } else if (hp->kd > 0) {
    const int64_t n_task = (int64_t)hp->kd * HYB_TDOT_CHUNKS;
    BBX_LAUNCH(hyb_dense_tdot_kernel,
                       dim3((unsigned)((n_task + 3) / 4)), dim3(256), 0,
                       h->stream, h->n, hp->kd, HYB_TDOT_CHUNKS,
                       hp->D.as<double>(), d_w, hp->d_part.as<double>(),
                       h->skip_flag);
    BBX_LAUNCH(hyb_dense_scatter_kernel, dim3((unsigned)hp->kd),
                       dim3(WAVE), 0, h->stream, hp->kd, HYB_TDOT_CHUNKS,
                       hp->dense_cols.as<int32_t>(), hp->d_part.as<double>(),
                       slab + (size_t)at * (size_t)h->p, h->skip_flag);
    BBX_HIP(hipGetLastError());
    at += 1;
  }
  hp->dw_for = nullptr;
  BBX_TRY(timer_end(h, 1));
  // the epilogue kernel adds the slabs in this order: B, S, D
  return launch_tdot_finalize(h, slab, at, d_sumw_part, ep, d_out);
}

// Returns 1 through *sum_done when the partial sums of the output were
// produced by the kernel itself (no separate reduction pass needed).
int launch_dot_tiled(bbx_design* h, const double* d_v,
                     const double* d_rowscale, double* d_t,
                     double* d_sum_part, int* sum_done, double* d_twt_part,
                     int* twt_done) {
  if (sum_done) *sum_done = 0;
  if (twt_done) *twt_done = 0;
  if (h->hybrid)
    return launch_dot_hybrid(h, d_v, d_rowscale, d_t, d_sum_part, sum_done,
                             d_twt_part, twt_done);
  TiledPair* tp = static_cast<TiledPair*>(h->tiled);
  const TiledMatrix& m = tp->x;
  const double* x = d_v + h->intercept;
  const double* x0 = h->intercept ? d_v : nullptr;
  if (m.G == 1) {
    double* fused = nullptr;
    int twt_off = 0;
    if (d_sum_part && m.n_panel <= NPART) {
      fused = d_sum_part;
      if (sum_done) *sum_done = 1;
      // <t, Omega t> partials ride along (both slots live in bbx_design::part)
      if (d_twt_part && twt_done && d_twt_part != d_sum_part) {
        twt_off = (int)(d_twt_part - d_sum_part);
        *twt_done = 1;
      }
    }
    hipEvent_t ea, eb;
    BBX_TRY(timer_arm(h, 0, &ea, &eb));
    return launch_tiled(h, m, x, part_slot(h, PS_C), x0, d_rowscale, d_t,
                        nullptr, fused, ea, eb, twt_off);
  }
  // G > 1: two kernels in the family, bracketed by a pair of record commands
  BBX_TRY(timer_begin(h, 0));
  BBX_TRY(launch_tiled(h, m, x, nullptr, nullptr, nullptr, nullptr,
                       m.slab.as<double>(), nullptr));
  BBX_LAUNCH(tiled_dot_finalize_kernel, dim3(1024), dim3(256), 0,
                     h->stream, m.R, m.G, m.slab.as<double>(),
                     part_slot(h, PS_C), x0, d_rowscale, d_t, nullptr);
  BBX_HIP(hipGetLastError());
  BBX_TRY(timer_end(h, 0));
  return BBX_OK;
}

bool tiled_fold_applies(const bbx_design* h) {
  // Default: on for designs of up to FOLD_MAX_ROWS rows.  The folded step costs
  // the X~ v kernel two n-vector passes (previous t in, new t out) plus the
  // direction work in front of its ring, and saves a P-vector launch with its
  // boundary: measured +3.5 % Gibbs it/s at 100k x 10k (935-974 against 895-950),
  // -1 % at 1M x 50k (LABNOTES.md R4.1).  BBX_CG_FOLD=0|1 and
  // bbx_design_set_cg_fold override.
  constexpr int64_t FOLD_MAX_ROWS = 250000;
  static const int env_on = getenv("BBX_CG_FOLD") ? atoi(getenv("BBX_CG_FOLD")) : -1;
  const bool on = h->cg_fold >= 0 ? h->cg_fold != 0
                  : env_on >= 0   ? env_on != 0
                                  : h->n <= FOLD_MAX_ROWS;
  if (!on || !h->sparse || h->format != BBX_FORMAT_TILED || h->hybrid || !h->tiled)
    return false;
  const TiledMatrix& m = static_cast<const TiledPair*>(h->tiled)->x;
  return m.G == 1 && m.n_panel <= NPART && !m.has_vals;
}

int launch_dot_tiled_fold(bbx_design* h, const DotFold& fa,
                          const double* d_rowscale, double* d_t,
                          double* d_sum_part, double* d_twt_part) {
  if (!tiled_fold_applies(h))
    return fail(BBX_ERR_STATE, "folded direction step does not apply");
  const TiledMatrix& m = static_cast<TiledPair*>(h->tiled)->x;
  h->n_dot += 1;
  const int twt_off = d_twt_part ? (int)(d_twt_part - d_sum_part) : 0;
  hipEvent_t ea, eb;
  BBX_TRY(timer_arm(h, 0, &ea, &eb));
  return launch_tiled(h, m, fa.sr + h->intercept, nullptr,
                      h->intercept ? fa.sr : nullptr, d_rowscale, d_t, nullptr,
                      d_sum_part, ea, eb, twt_off, nullptr, nullptr, &fa);
}

int launch_tdot_tiled(bbx_design* h, const double* d_w,
                      const double* d_sumw_part, const TdotEpilogue& ep,
                      double* d_out) {
  if (h->hybrid) return launch_tdot_hybrid(h, d_w, d_sumw_part, ep, d_out);
  TiledPair* tp = static_cast<TiledPair*>(h->tiled);
  const TiledMatrix& m = tp->xt;
  hipEvent_t ea, eb;
  BBX_TRY(timer_arm(h, 1, &ea, &eb));
  BBX_TRY(launch_tiled(h, m, d_w, nullptr, nullptr, nullptr, nullptr,
                       m.slab.as<double>(), nullptr, ea, eb));
  // the epilogue kernel adds the G partial slabs in group order
  return launch_tdot_finalize(h, m.slab.as<double>(), m.G, d_sumw_part, ep,
                              d_out);
}

// ---- batched products (K right-hand sides, one pass over the id stream)

// t_c[r] = rowscale_c[r] * (c_c + sum_g slab[g][r][c])   (dot with G > 1), plus
// the per-chain partials of sum(t_c) and <t_c, Omega_c t_c>.  Grid = NPART.
template <int K>
__global__ __launch_bounds__(256) void tiled_dot_finalize_k_kernel(
    int64_t R, int G, const double* __restrict__ slab,
    const double* __restrict__ c_part, const double* x0_ptr,
    ChainPtrs rowscale, ChainOut out, int out_stride, int part_stride,
    double* __restrict__ sum_part, int twt_off,
    const double* __restrict__ addend) {
  __shared__ double s_c[K];
  __shared__ double s_w[2 * K][256 / WAVE];
  if (threadIdx.x < K) {
    double c = x0_ptr ? x0_ptr[threadIdx.x] : 0.;
    if (c_part) {
      double cs = 0.;
      for (int k = 0; k < NPART; ++k) cs += c_part[threadIdx.x * part_stride + k];
      c -= cs;
    }
    s_c[threadIdx.x] = c;
  }
  __syncthreads();
  double tsum[K], t2sum[K];
#pragma unroll
  for (int c = 0; c < K; ++c) tsum[c] = t2sum[c] = 0.;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < R;
       r += (int64_t)gridDim.x * 256) {
#pragma unroll
    for (int c = 0; c < K; ++c) {
      double a = 0.;
      for (int g = 0; g < G; ++g) a += slab[((int64_t)g * R + r) * K + c];
      if (addend) a += addend[r * K + c];
      const double t = s_c[c] + a;
      double v = t;
      if (rowscale.p[c]) v *= rowscale.p[c][r];
      out.p[c][r * out_stride] = v;
      tsum[c] += v;
      t2sum[c] += v * t;
    }
  }
  if (!sum_part) return;
#pragma unroll
  for (int c = 0; c < K; ++c) {
    const double a = wave_allsum(tsum[c]), b = wave_allsum(t2sum[c]);
    if ((threadIdx.x & (WAVE - 1)) == 0) {
      s_w[2 * c][threadIdx.x / WAVE] = a;
      s_w[2 * c + 1][threadIdx.x / WAVE] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double a = 0., b = 0.;
    for (int k = 0; k < 256 / WAVE; ++k) {
      a += s_w[2 * threadIdx.x][k];
      b += s_w[2 * threadIdx.x + 1][k];
    }
    sum_part[threadIdx.x * part_stride + blockIdx.x] = a;
    if (twt_off) sum_part[threadIdx.x * part_stride + twt_off + blockIdx.x] = b;
  }
}

int launch_dot_tiled_k(bbx_design* h, int K, const double* d_v,
                       const double* d_c_part, const TiledBatchArgs& ba,
                       double* d_sum_part, int twt_off) {
  const TiledPair* tp = tiled_pair_for(h, K);
  if (!tp || K < 2) return fail(BBX_ERR_STATE, "batched layout not built");
  const TiledMatrix& m = tp->x;
  const double* x = d_v + (size_t)h->intercept * K;
  const double* x0 = h->intercept ? d_v : nullptr;
  h->n_dot += 1;
  // mixed design kept split: the dense block's product first, the value-free
  // kernel's epilogue adds it
  HybridParts* hp = static_cast<HybridParts*>(h->hybrid);
  const bool split = hp && hp->split_k[K == 2 ? 0 : 1];
  const double* addend = nullptr;
  if (split && (hp->kd > 0 || hp->rest_nnz > 0)) {
    const double* rest_slab = nullptr;
    int G_rest = 0;
    if (hp->rest_nnz > 0) {
      const TiledMatrix& ms = hp->rest_k->x;
      TiledBatchArgs slab_rest = ba;
      slab_rest.out = ChainOut{};
      BBX_TRY(launch_tiled(h, ms, x, nullptr, nullptr, nullptr, nullptr,
                           ms.slab.as<double>(), nullptr, nullptr, nullptr, 0,
                           &slab_rest));
      rest_slab = ms.slab.as<double>();
      G_rest = ms.G;
    }
    const size_t lds = sizeof(double) * (size_t)(hp->kd * K + 1);
#define BBX_ADDEND_K(KK)                                                       \
  BBX_LAUNCH(hyb_addend_k_kernel<KK>, dim3(1024), dim3(256), lds,      \
                     h->stream, h->n, hp->kd, h->intercept, hp->D.as<double>(), \
                     hp->dense_cols.as<int32_t>(), d_v, rest_slab, G_rest,     \
                     hp->addend_k.as<double>(), h->skip_flag)
    if (K == 2) BBX_ADDEND_K(2); else BBX_ADDEND_K(4);
#undef BBX_ADDEND_K
    BBX_HIP(hipGetLastError());
    addend = hp->addend_k.as<double>();
  }
  if (m.G == 1 && m.n_panel <= NPART) {
    hipEvent_t ea, eb;
    BBX_TRY(timer_arm(h, 0, &ea, &eb));
    return launch_tiled(h, m, x, d_c_part, x0, nullptr, nullptr, nullptr,
                        d_sum_part, ea, eb, twt_off, &ba, addend);
  }
  // several column groups (or more panels than partial slots): slabs, then
  // the finalize kernel
  BBX_TRY(timer_begin(h, 0));
  TiledBatchArgs slab_only = ba;
  slab_only.out = ChainOut{};
  BBX_TRY(launch_tiled(h, m, x, nullptr, nullptr, nullptr, nullptr,
                       m.slab.as<double>(), nullptr, nullptr, nullptr, 0,
                       &slab_only));
#define BBX_DOT_FIN(KK)                                                        \
  BBX_LAUNCH(tiled_dot_finalize_k_kernel<KK>, dim3(NPART), dim3(256),  \
                     0, h->stream, m.R, m.G, m.slab.as<double>(), d_c_part,    \
                     x0, ba.rowscale, ba.out, ba.out_stride, ba.part_stride,   \
                     d_sum_part, twt_off, addend)
  if (K == 2) BBX_DOT_FIN(2); else BBX_DOT_FIN(4);
#undef BBX_DOT_FIN
  BBX_HIP(hipGetLastError());
  return timer_end(h, 0);
}

int launch_tdot_tiled_k(bbx_design* h, int K, const double* d_w,
                        const double** slab, int* G) {
  const TiledPair* tp = tiled_pair_for(h, K);
  if (!tp || K < 2) return fail(BBX_ERR_STATE, "batched layout not built");
  const TiledMatrix& m = tp->xt;
  h->n_tdot += 1;
  hipEvent_t ea, eb;
  BBX_TRY(timer_arm(h, 1, &ea, &eb));
  TiledBatchArgs none;
  HybridParts* hp = static_cast<HybridParts*>(h->hybrid);
  const int ks = K == 2 ? 0 : 1;
  if (hp && hp->split_k[ks]) {
    // mixed design kept split: B's slabs, then one more slab for D^T W
    double* sl = hp->slab_k[ks].as<double>();
    BBX_TRY(launch_tiled(h, m, d_w, nullptr, nullptr, nullptr, nullptr, sl,
                         nullptr, ea, eb, 0, &none));
    int at = m.G;
    if (hp->rest_nnz > 0) {
      const TiledMatrix& ms = hp->rest_k->xt;
      BBX_TRY(launch_tiled(h, ms, d_w, nullptr, nullptr, nullptr, nullptr,
                           sl + (size_t)at * (size_t)h->p * (size_t)K, nullptr,
                           nullptr, nullptr, 0, &none));
      at += ms.G;
    }
    if (hp->kd > 0) {
      const int64_t n_task = (int64_t)hp->kd * HYB_TDOT_CHUNKS;
      double* row = sl + (size_t)at * (size_t)h->p * (size_t)K;
#define BBX_DENSE_TDOT_K(KK)                                                   \
  do {                                                                         \
    BBX_LAUNCH(hyb_dense_tdot_k_kernel<KK>,                            \
                       dim3((unsigned)((n_task + 3) / 4)), dim3(256), 0,       \
                       h->stream, h->n, hp->kd, HYB_TDOT_CHUNKS,               \
                       hp->D.as<double>(), d_w, hp->d_part_k.as<double>(),     \
                       h->skip_flag);                                          \
    BBX_LAUNCH(hyb_dense_scatter_k_kernel<KK>, dim3((unsigned)hp->kd), \
                       dim3(WAVE), 0, h->stream, hp->kd, HYB_TDOT_CHUNKS,      \
                       hp->dense_cols.as<int32_t>(),                           \
                       hp->d_part_k.as<double>(), row, h->skip_flag);          \
  } while (0)
      if (K == 2) BBX_DENSE_TDOT_K(2); else BBX_DENSE_TDOT_K(4);
#undef BBX_DENSE_TDOT_K
      BBX_HIP(hipGetLastError());
      at += 1;
    }
    *slab = sl;
    *G = at;
    return BBX_OK;
  }
  BBX_TRY(launch_tiled(h, m, d_w, nullptr, nullptr, nullptr, nullptr,
                       m.slab.as<double>(), nullptr, ea, eb, 0, &none));
  *slab = m.slab.as<double>();
  *G = m.G;
  return BBX_OK;
}

// What ONE batched launch of each product moves (the timed kernels): the id
// stream of the K-layout + K vectors in + K vectors (dot) or G slabs of K
// columns (Tdot) out.
int tiled_batch_bytes(const bbx_design* h, int K, int64_t* dot_bytes,
                      int64_t* tdot_bytes) {
  const TiledPair* tp = tiled_pair_for(h, K);
  if (!tp) return fail(BBX_ERR_STATE, "batched layout not built");
  *dot_bytes = tp->x.stream_bytes() + 8 * (int64_t)K * (h->P + h->n) +
               (tp->x.G > 1 ? 16 * (int64_t)K * tp->x.G * h->n : 0);
  *tdot_bytes = tp->xt.stream_bytes() + 8 * (int64_t)K * h->n +
                8 * (int64_t)K * tp->xt.G * h->p;
  const HybridParts* hp = static_cast<const HybridParts*>(h->hybrid);
  if (hp && hp->split_k[K == 2 ? 0 : 1]) {
    // the dense block once per product, the addend written and read back
    const int64_t dense = 8 * (int64_t)hp->kd * h->n;
    *dot_bytes += dense + 16 * (int64_t)K * h->n;
    *tdot_bytes += dense + 8 * (int64_t)K * h->n;
    if (hp->rest_k) {
      *dot_bytes += hp->rest_k->x.stream_bytes() + 16 * (int64_t)K * hp->rest_k->x.G * h->n;
      *tdot_bytes += hp->rest_k->xt.stream_bytes() + 8 * (int64_t)K * hp->rest_k->xt.G * h->p;
    }
  }
  return BBX_OK;
}

int tiled_matvec_bytes(const bbx_design* h, int64_t* dot_bytes,
                       int64_t* tdot_bytes, bool timed_only) {
  if (h->hybrid) {
    // every part's stream once, the dense block once, the rest part's slabs
    // and the addend written and read back, vectors in and out; the timers
    // bracket everything but the final epilogue kernel of the Tdot
    const HybridParts* hp = static_cast<const HybridParts*>(h->hybrid);
    const int64_t dense = 8 * (int64_t)hp->kd * h->n;
    int64_t db = hp->ones.x.stream_bytes() + dense + 8 * (h->P + h->n) +
                 16 * h->n;
    int64_t tb = hp->ones.xt.stream_bytes() + dense + 8 * h->n +
                 8 * (int64_t)hp->n_slab * h->p;
    if (hp->rest_nnz > 0) {
      db += hp->rest.x.stream_bytes() + 16 * (int64_t)hp->rest.x.G * h->n + 8 * h->P;
      tb += hp->rest.xt.stream_bytes() + 8 * h->n;
    }
    if (!timed_only) tb += 8 * (int64_t)hp->n_slab * h->p + 8 * h->P;
    *dot_bytes = db;
    *tdot_bytes = tb;
    return BBX_OK;
  }
  const TiledPair* tp = static_cast<const TiledPair*>(h->tiled);
  if (!tp) return fail(BBX_ERR_STATE, "tiled format not built");
  // Whole product: bytes of the format actually read + vector in + vector out
  // + the partial slabs written by the main kernel and read back by the
  // epilogue kernel.  Dot with G == 1 is a single kernel; with G > 1 the timer
  // brackets both kernels, so timed == whole.  Tdot: the timer stamps the main
  // kernel only (ids + w in + G slabs out); the epilogue kernel (slab read,
  // P-vector out) is outside it.
  *dot_bytes = tp->x.stream_bytes() + 8 * (h->P + h->n) +
               (tp->x.G > 1 ? 16 * tp->x.G * h->n : 0);
  // inside the CG loop the X~ v kernel carries the direction step (DotFold):
  // the previous t in and the new unscaled t out (n-vectors), per coordinate
  // r, p, d in and p out -- that is the kernel the timers stamp
  if (timed_only && tiled_fold_applies(h)) *dot_bytes += 16 * h->n + 4 * 8 * h->P;
  if (timed_only)
    *tdot_bytes = tp->xt.stream_bytes() + 8 * h->n +
                  8 * (int64_t)tp->xt.G * h->p;
  else
    *tdot_bytes = tp->xt.stream_bytes() + 8 * (h->n + h->P) +
                  16 * (int64_t)tp->xt.G * h->p;
  return BBX_OK;
}

// What the TIMED kernels move that a reader of the matrix cannot do without:
// the ids of the stored entries at the layout's rate (2 bytes, or 1.6 in groups
// of five; + 8 for a stored value), the row ids of the slices, the vector in and
// the output -- no padding of the steps, no schedules.  *pad_* = the share of the
// id (and value) stream that is padding.
int tiled_useful_bytes(const bbx_design* h, int64_t* dot_bytes,
                       int64_t* tdot_bytes, double* pad_dot, double* pad_tdot) {
  const TiledPair* tp = static_cast<const TiledPair*>(h->tiled);
  if (h->hybrid || !tp) return fail(BBX_ERR_STATE, "plain tiled designs only");
  auto ids = [](const TiledMatrix& m, double* pad) {
    const double per_entry = (m.packed ? 1.6 : 2.0) + (m.has_vals ? 8. : 0.);
    const double useful = per_entry * (double)m.nnz;
    const double moved = (double)m.n_quad * 64. * 16. * (m.has_vals ? 5. : 1.);
    if (pad) *pad = moved > 0. ? 1. - useful / moved : 0.;
    return (int64_t)useful + (int64_t)m.n_slice * 256;
  };
  *dot_bytes = ids(tp->x, pad_dot) + 8 * (h->P + h->n) +
               (tp->x.G > 1 ? 16 * tp->x.G * h->n : 0);
  *tdot_bytes = ids(tp->xt, pad_tdot) + 8 * h->n + 8 * (int64_t)tp->xt.G * h->p;
  return BBX_OK;
}

int tiled_hybrid_info(const bbx_design* h, int64_t* ones_nnz,
                      int64_t* rest_nnz, int64_t* dense_nnz, int* kd) {
  const HybridParts* hp = static_cast<const HybridParts*>(h->hybrid);
  if (ones_nnz) *ones_nnz = hp ? hp->ones_nnz : 0;
  if (rest_nnz) *rest_nnz = hp ? hp->rest_nnz : 0;
  if (dense_nnz) *dense_nnz = hp ? hp->dense_nnz : 0;
  if (kd) *kd = hp ? hp->kd : 0;
  return hp ? 1 : 0;
}

int64_t tiled_storage_bytes(const bbx_design* h) {
  if (h->hybrid) {
    const HybridParts* hp = static_cast<const HybridParts*>(h->hybrid);
    int64_t b = hp->ones.x.stream_bytes() + hp->ones.xt.stream_bytes() +
                8 * (int64_t)hp->kd * h->n;
    if (hp->rest_nnz > 0)
      b += hp->rest.x.stream_bytes() + hp->rest.xt.stream_bytes();
    return b;
  }
  const TiledPair* tp = static_cast<const TiledPair*>(h->tiled);
  if (!tp) return 0;
  return tp->x.stream_bytes() + tp->xt.stream_bytes();
}

int tiled_describe(const bbx_design* h, int which, int* W, int* n_block,
                   int* PR, int* G, int64_t* n_quad, int64_t* n_slice,
                   int* packed) {
  if (which < 0 || which > 7)
    return fail(BBX_ERR_INVALID, "which must be 0 ... 7");
  const TiledPair* tp = static_cast<const TiledPair*>(h->tiled);
  // mixed designs: the value-free part (0, 1) and the valued rest (6, 7)
  if (h->hybrid) {
    const HybridParts* hp = static_cast<const HybridParts*>(h->hybrid);
    tp = (which >= 6) ? (hp->rest_nnz > 0 ? &hp->rest : nullptr) : &hp->ones;
    if (which >= 6) which -= 6;
  } else if (which >= 6) {
    return fail(BBX_ERR_INVALID,
                "which = 6, 7 name the valued rest of a MIXED design");
  }
  if (!tp) return fail(BBX_ERR_STATE, "tiled format not built");
  if (which >= 2) {  // 2, 3: the K = 2 layout; 4, 5: the K = 4 layout
    tp = tiled_pair_for(h, which < 4 ? 2 : 4);
    if (!tp) return fail(BBX_ERR_STATE, "batched layout not built");
  }
  const TiledMatrix& m = (which & 1) == 0 ? tp->x : tp->xt;
  if (W) *W = m.W;
  if (n_block) *n_block = m.n_block;
  if (PR) *PR = m.PR;
  if (G) *G = m.G;
  if (n_quad) *n_quad = m.n_quad;
  if (n_slice) *n_slice = m.n_slice;
  if (packed) *packed = m.packed ? 1 : 0;
  return BBX_OK;
}

}  // namespace bbx
