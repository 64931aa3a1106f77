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
//     a multiple of 4 entries with an id that points at a 0.0 in LDS;
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
// nnz: 2 bytes per entry), adds lane-private sums into LDS accumulators, and
// writes the panel once.  No atomics: every sum has a fixed order.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>

#include <hip/hip_ext.h>

#include "common.hpp"

// -DBBX_TILED_INSTRUMENT=1 compiles the per-wave phase timers in (they cost
// ~8 SGPRs, which the production kernel has no room for); BBX_TILED_DEBUG=N
// then prints the breakdown of launches N and N+1.
#ifndef BBX_TILED_INSTRUMENT
#define BBX_TILED_INSTRUMENT 0
#endif
// Register ring of the value-free kernel: RING slots of BATCH steps each.
// With non-temporal id loads the kernel is insensitive to the bytes a wave has
// in flight (even ONE 1 KiB step in flight per wave runs as fast); shallow and
// fine-grained wins by a little.  Measured at 1M x 50k, X v / X^T w in us:
// 4x2: 48.6 / 50.8   3x2: 47.6 / 50.0   4x1: 46.8 / 49.5   3x1: 46.7 / 49.3
// 2x1: 46.7 / 49.7   6x1: 47.4 / 50.1   (100k x 10k Gibbs: 4x2 614, 3x1 675 it/s)
#ifndef BBX_RING_BIN
#define BBX_RING_BIN 3
#endif

namespace bbx {

// Workgroup geometry.  Default: one 1024-thread workgroup owns a CU's LDS.
// -DBBX_TILE_THREADS=512 -DBBX_TILE_LDS_KB=80 -DBBX_TILE_W_MAX=7168 builds the
// "two half-width workgroups per CU" variant measured in DESIGN.md 3.1.
#ifndef BBX_TILE_THREADS
#define BBX_TILE_THREADS 1024
#endif
#ifndef BBX_TILE_LDS_KB
#define BBX_TILE_LDS_KB 160
#endif
#ifndef BBX_TILE_W_MAX
#define BBX_TILE_W_MAX 16128
#endif
constexpr int TILE_W_MAX = BBX_TILE_W_MAX;  // doubles of the vector slice in LDS
constexpr int TILE_PR_MAX = 4096;  // row accumulators in LDS
constexpr int TILE_THREADS = BBX_TILE_THREADS;
constexpr int TILE_LDS_BYTES = BBX_TILE_LDS_KB * 1024;
constexpr int TILE_WG_PER_CU = (160 / BBX_TILE_LDS_KB);
constexpr int TILE_WG_PER_ROUND = 256 * TILE_WG_PER_CU;
constexpr int TILE_WAVES = TILE_THREADS / WAVE;
constexpr uint16_t NO_ROW = 0xFFFF;

// Set-up only (the kernel derives the column block arithmetically).
struct TileDesc {
  int32_t col_block;
  int32_t slice_begin;
  int32_t slice_end;
  int32_t pad;
};

constexpr int SLICE_ROWS = 2 * WAVE;  // two rows per lane

struct SliceMeta {
  uint32_t first_quad;  // offset into the id stream in units of 64 uint4
  uint32_t n_quad;      // steps: 4 entries of row A + 4 of row B per lane
};

// One step of a wave's precomputed schedule: BATCH consecutive quads of one
// slice.  The schedule of every (workgroup, wave) is laid out in processing
// order, so the kernel's issue cursor is a single scalar index.
struct BatchDesc {
  uint32_t quad0;     // first step (units of 64 uint4 in the id stream)
  uint32_t row_slot;  // slice * 64: where the slice's row-id pairs start
  uint32_t info;      // bits 0-3 count, 8 last-of-slice, 9 tile-first, 10 end
  uint32_t pad;
};
constexpr uint32_t BD_LAST = 1u << 8;
constexpr uint32_t BD_TILE_FIRST = 1u << 9;
constexpr uint32_t BD_END = 1u << 10;
#ifndef BBX_BATCH_BIN
#define BBX_BATCH_BIN 1
#endif
constexpr int BATCH_BIN = BBX_BATCH_BIN;  // steps per ring slot, value-free
constexpr int BATCH_VAL = 1;   // steps per ring slot when values are stored

// One orientation (X or X^T) in tiled form, device resident.
struct TiledMatrix {
  int64_t R = 0, C = 0, nnz = 0;
  int W = 0, n_block = 0, PR = 0, n_panel = 0, G = 0;
  bool has_vals = false;
  bool packed = false;  // value-free ids as 14-bit base + 4 x 12-bit deltas
  int64_t n_slice = 0, n_quad = 0, n_tile = 0;
  DevMem ids;        // uint4[n_quad * 64]
  DevMem vals;       // double[n_quad * 64 * 8] when has_vals
  DevMem descs;      // BatchDesc[n_desc]: per-wave schedules
  DevMem wave_desc;  // int32[n_panel * G * 16]: first descriptor of each wave
  int desc_stride = 0;  // > 0: wave k's schedule starts at k * desc_stride
  int64_t n_desc = 0;
  DevMem rowids;     // uint32[n_slice * 64]: panel-local rows A | B << 16
  DevMem folds;      // FoldDesc[n_fold]
  DevMem panel_fold; // int32[n_panel + 1]
  int n_extra = 0;   // extra accumulators per panel (row splitting)
  int split_T = 0;   // smallest split threshold used by any panel (0 = none)
  DevMem slab;       // double[G * R] partial sums when G > 1 (or Tdot)
  int64_t stream_bytes() const {
    return (int64_t)n_quad * 64 * 16 * (has_vals ? 5 : 1) +
           (int64_t)n_slice * 256 + (int64_t)n_desc * (int64_t)sizeof(BatchDesc);
  }
};

// Row r of a panel was split: acc[r] += acc[first .. first+count) at the end.
struct FoldDesc {
  uint16_t row, first, count, pad;
};

struct TiledPair {
  TiledMatrix x, xt;
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

// Packed value-free step: per row one 64-bit group = 14-bit block-local id of
// the first entry + four 12-bit forward deltas (5 entries in 8 bytes instead
// of 4).  A zero delta marks "no further entry" (ids ascend strictly inside a
// group; the builder starts a new group at a duplicate or at a gap > 4095), and
// an empty group has base id W, where LDS holds 0.0.
__device__ __forceinline__ void packed_row(const double* __restrict__ xs,
                                           unsigned lo, unsigned hi,
                                           unsigned zero_slot, double& s0,
                                           double& s1) {
  const unsigned long long g = (unsigned long long)lo |
                               ((unsigned long long)hi << 32);
  const unsigned i0 = lo & 0x3FFFu;
  const unsigned d1 = (lo >> 14) & 0xFFFu;
  const unsigned d2 = (unsigned)(g >> 26) & 0xFFFu;
  const unsigned d3 = (hi >> 6) & 0xFFFu;
  const unsigned d4 = (hi >> 18) & 0xFFFu;
  const unsigned i1 = i0 + d1, i2 = i1 + d2, i3 = i2 + d3, i4 = i3 + d4;
  s0 += xs[i0] + xs[d2 ? i2 : zero_slot] + xs[d4 ? i4 : zero_slot];
  s1 += xs[d1 ? i1 : zero_slot] + xs[d3 ? i3 : zero_slot];
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

template <bool VALS, bool PACK, bool WIDE>
__global__ __launch_bounds__(TILE_THREADS, 4) void tiled_spmv_kernel(
    int64_t R, int64_t C, int W, int PR, int G, int blocks_per_group,
    const int32_t* __restrict__ wave_desc, int desc_stride,
    const BatchDesc* __restrict__ descs, const uint32_t* __restrict__ rowids,
    const uint4* __restrict__ ids, const double* __restrict__ vals,
    const double* __restrict__ x,
    // epilogue (direct mode, G == 1 and out != nullptr):
    //   out[r] = rowscale[r] * (c0 - sum(c_part) + acc)
    const double* __restrict__ c_part, const double* x0_ptr,
    const double* __restrict__ rowscale, double* __restrict__ out,
    double* __restrict__ slab, int n_acc,
    const int32_t* __restrict__ panel_fold, const FoldDesc* __restrict__ folds,
    double* __restrict__ out_sum_part, int ablate,
    unsigned long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* xs = lds;               // W + 8 doubles; xs[W] == 0 (padding target)
  double* acc = lds + (W + 8);    // n_acc = PR + extra doubles
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
  const int panel = blockIdx.x / G;
  const int group = blockIdx.x - panel * G;
  const int64_t row0 = (int64_t)panel * PR;
  const int rows_here = (int)((R - row0 < PR) ? (R - row0) : PR);

  for (int r = tid; r < n_acc; r += TILE_THREADS) acc[r] = 0.;
  if (tid < 8) xs[W + tid] = 0.;

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
                ? (int)(blockIdx.x * TILE_WAVES + wave) * desc_stride
                : wave_desc[blockIdx.x * TILE_WAVES + wave];
  int pos = 0;
  const uint4* __restrict__ desc4 = reinterpret_cast<const uint4*>(descs);
  uint4 dcur = desc4[blk + lane];
  uint4 dnxt = desc4[blk + WAVE + lane];
  // Retire every compiler-visible load before the ring starts (see ISSUE).
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
  v4u e[RING][BATCH];
  v2d ev[RING][BATCH][NV];
  unsigned rid[RING];
  unsigned info[RING];
  double a0 = 0., a1 = 0., b0 = 0., b1 = 0.;
  // BBX_TILED_DEBUG: per-wave cycle stamps (start, time in tile switches,
  // end of the stream loop, end of the kernel); dbg is null in production.
  // 32-bit tick counts (durations only: s_memtime bases differ across XCDs)
  unsigned t_start = 0, t_switch = 0, t_loop = 0, t_skew = 0;
  if (!BBX_TILED_INSTRUMENT) dbg = nullptr;  // folds every timer away

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
                if (j < W) *reinterpret_cast<v2d*>(xs + j) = fp[u];  // W even
              }
            } else {
#pragma unroll
              for (int u = 0; u < FILL_PAIRS; ++u) {
                const int j0 = tid + (2 * u) * TILE_THREADS;
                const int j1 = tid + (2 * u + 1) * TILE_THREADS;
                if (j0 < W) xs[j0] = fp[u].x;
                if (j1 < W) xs[j1] = fp[u].y;
              }
            }
            if (!(ablate & 4)) __syncthreads();
            if (dbg) t_switch += (unsigned)__builtin_amdgcn_s_memtime() - t_sw0;
          }
          const int cntk = (int)(inf & 15u);
          if (cntk > 0) {
            BBX_WAIT(k);
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
              if (u < cntk) {
                if (ablate & 1)
                  a0 += (double)(e[k][u].x ^ e[k][u].y ^ e[k][u].z ^ e[k][u].w);
                else if (PACK) {
                  packed_row(xs, e[k][u].x, e[k][u].y, (unsigned)W, a0, a1);
                  packed_row(xs, e[k][u].z, e[k][u].w, (unsigned)W, b0, b1);
                } else
                  step_accumulate<VALS>(xs, e[k][u], ev[k][u], a0, a1, b0, b1);
              }
            if (inf & BD_LAST) {
              const unsigned rr = rid[k];
              const unsigned ra = rr & 0xFFFFu, rb = rr >> 16;
              if (ra != NO_ROW) acc[ra] += a0 + a1;
              if (rb != NO_ROW) acc[rb] += b0 + b1;
              a0 = a1 = b0 = b1 = 0.;
            }
          }
          BBX_ISSUE(k);
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
  // Everything the epilogue reads from memory and that does not depend on the
  // accumulators is requested NOW, before the wave joins the final barrier:
  // the round trip (the row scale is an HBM-resident n-vector) overlaps the
  // wait for the slowest wave instead of following it.
  constexpr int EPI_UNROLL = TILE_PR_MAX / TILE_THREADS;
  double rs_pre[EPI_UNROLL];
  double cp_pre[NPART / WAVE];
  double x0_pre = 0.;
  if (out) {
#pragma unroll
    for (int u = 0; u < EPI_UNROLL; ++u) {
      const int r = tid + u * TILE_THREADS;
      rs_pre[u] = (rowscale && r < rows_here) ? rowscale[row0 + r] : 1.;
    }
#pragma unroll
    for (int k = 0; k < NPART / WAVE; ++k)
      cp_pre[k] = c_part ? c_part[lane + k * WAVE] : 0.;
    if (x0_ptr) x0_pre = *x0_ptr;
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
    if (tid < WAVE) {
      double cs = 0.;
      if (c_part) {
#pragma unroll
        for (int k = 0; k < NPART / WAVE; ++k) cs += cp_pre[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cs += __shfl_down(cs, off, WAVE);
      }
      if (tid == 0) xs[0] = x0_pre - cs;
    }
    __syncthreads();
    const double c = xs[0];
    double tsum = 0.;
#pragma unroll
    for (int u = 0; u < EPI_UNROLL; ++u) {
      const int r = tid + u * TILE_THREADS;
      if (r < rows_here) {
        double v = c + acc[r];
        if (rowscale) v *= rs_pre[u];
        out[row0 + r] = v;
        tsum += v;
      }
    }
    if (out_sum_part) {
      // partial sum of this panel's outputs (feeds the intercept / centring
      // terms of the following Tdot); fixed order: lanes, then waves
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
        tsum += __shfl_down(tsum, off, WAVE);
      __syncthreads();
      if (lane == 0) xs[wave] = tsum;
      __syncthreads();
      if (tid == 0) {
        double tot = 0.;
        for (int wv = 0; wv < TILE_WAVES; ++wv) tot += xs[wv];
        out_sum_part[blockIdx.x] = tot;
      }
      // consumers add NPART slots: the first workgroup clears the unused ones
      if (blockIdx.x == 0 && (int)gridDim.x + tid < NPART)
        out_sum_part[gridDim.x + tid] = 0.;
    }
  } else {
    double* dst = slab + (int64_t)group * R + row0;
    for (int r = tid; r < rows_here; r += TILE_THREADS) dst[r] = acc[r];
  }
  if (dbg && lane == 0) {
    unsigned long long* o = dbg + ((size_t)blockIdx.x * TILE_WAVES + wave) * 4;
    o[0] = (unsigned)__builtin_amdgcn_s_memtime() - t_start;  // whole wave
    o[1] = t_loop;    // stream loop incl. tile switches
    o[2] = t_switch;  // inside tile switches
    o[3] = t_skew;    // of which: arrival -> every wave arrived
  }
}

// out[r] = rowscale[r] * (c + sum_g slab[g][r])   (dot with G > 1)
__global__ __launch_bounds__(256) void tiled_dot_finalize_kernel(
    int64_t R, int G, const double* __restrict__ slab,
    const double* __restrict__ c_part, const double* x0_ptr,
    const double* __restrict__ rowscale, double* __restrict__ out) {
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
    if (rowscale) v *= rowscale[r];
    out[r] = v;
  }
}

// ----------------------------------------------------------------- builder

struct PanelBuild {
  std::vector<uint4> ids;
  std::vector<double> vals;
  std::vector<SliceMeta> slices;  // first_quad local to the panel
  std::vector<uint32_t> rowids;
  std::vector<TileDesc> tiles;    // slice ids local to the panel
  std::vector<int32_t> group_tile_count;
  std::vector<FoldDesc> folds;    // split rows of this panel
  int split_T = 0;
  std::vector<BatchDesc> descs;   // quad0/row_slot local to the panel
  std::vector<int32_t> wave_desc; // [G * TILE_WAVES] local start of each wave
  // schedule statistics (BBX_TILED_STATS): per workgroup, in batches
  std::vector<int64_t> wg_critical;  // sum over tiles of the busiest wave
  std::vector<int64_t> wg_total;     // all waves, all tiles
  int64_t dup_quads = 0;             // quads re-loaded to fill a batch
};

// Per-wave schedules of one panel: for every workgroup (group of column
// blocks) and wave, the batches of its slices in processing order.  Inside a
// tile the slices (sorted by decreasing length) go one by one to the wave with
// the least work so far (longest-processing-time rule), which keeps the 16
// waves of a workgroup within one slice of each other at the tile barrier.
static void build_schedules(PanelBuild& pb, int G, int batch) {
  pb.wave_desc.assign((size_t)G * TILE_WAVES, 0);
  size_t tile_cursor = 0;
  // The id stream is re-laid in the order it will be READ: workgroup, wave,
  // tile, slice.  Every wave then walks one contiguous region of HBM front to
  // back (consecutive 1 KiB loads, DRAM-page friendly) instead of hopping
  // between the slices the dealing happened to give it.
  const bool has_vals = !pb.vals.empty();
  std::vector<uint4> new_ids;
  std::vector<double> new_vals;
  std::vector<uint32_t> new_rowids;
  std::vector<SliceMeta> new_slices;
  new_ids.reserve(pb.ids.size());
  new_rowids.reserve(pb.rowids.size());
  new_slices.reserve(pb.slices.size());
  if (has_vals) new_vals.reserve(pb.vals.size());
  for (int g = 0; g < G; ++g) {
    const size_t t0 = tile_cursor, t1 = tile_cursor + pb.group_tile_count[g];
    tile_cursor = t1;
    // tile_deal[t - t0][w] = slices of tile t handled by wave w
    std::vector<std::vector<std::vector<int>>> tile_deal(t1 - t0);
    for (size_t t = t0; t < t1; ++t) {
      const TileDesc& td = pb.tiles[t];
      std::vector<std::vector<int>>& dl = tile_deal[t - t0];
      dl.assign(TILE_WAVES, std::vector<int>());
      int64_t load[TILE_WAVES];
      for (int w = 0; w < TILE_WAVES; ++w) load[w] = 0;
      for (int sl = td.slice_begin; sl < td.slice_end; ++sl) {
        // rotate the tie-break with the tile index so that no wave is
        // systematically first
        int best = (int)((t + (size_t)sl) % TILE_WAVES);
        for (int k = 0; k < TILE_WAVES; ++k) {
          const int w = (int)((t + (size_t)k) % TILE_WAVES);
          if (load[w] < load[best]) best = w;
        }
        dl[best].push_back(sl);
        // cost: steps plus a per-slice overhead (row ids, flush)
        load[best] += (int64_t)pb.slices[(size_t)sl].n_quad + 2;
      }
    }
    {
      int64_t crit = 0, total = 0;
      for (size_t t = t0; t < t1; ++t) {
        int64_t worst = 0;
        for (int w = 0; w < TILE_WAVES; ++w) {
          int64_t nb = 0;
          for (int sl : tile_deal[t - t0][w]) {
            const int64_t nq = pb.slices[(size_t)sl].n_quad;
            const int64_t b = (nq + batch - 1) / batch;
            nb += b;
            pb.dup_quads += b * batch - nq;
          }
          worst = std::max(worst, nb);
          total += nb;
        }
        crit += worst;
      }
      pb.wg_critical.push_back(crit);
      pb.wg_total.push_back(total);
    }
    for (int w = 0; w < TILE_WAVES; ++w) {
      pb.wave_desc[(size_t)g * TILE_WAVES + w] = (int32_t)pb.descs.size();
      for (size_t t = t0; t < t1; ++t) {
        bool first = true;
        for (int sl : tile_deal[t - t0][w]) {
          const SliceMeta& old = pb.slices[(size_t)sl];
          // move the slice to the end of the re-laid stream
          SliceMeta sm;
          sm.first_quad = (uint32_t)(new_ids.size() / WAVE);
          sm.n_quad = old.n_quad;
          const uint32_t new_sl = (uint32_t)new_slices.size();
          new_slices.push_back(sm);
          const size_t src = (size_t)old.first_quad * WAVE;
          const size_t cnt = (size_t)old.n_quad * WAVE;
          new_ids.insert(new_ids.end(), pb.ids.begin() + src,
                         pb.ids.begin() + src + cnt);
          if (has_vals)
            new_vals.insert(new_vals.end(), pb.vals.begin() + src * 8,
                            pb.vals.begin() + (src + cnt) * 8);
          new_rowids.insert(new_rowids.end(),
                            pb.rowids.begin() + (size_t)sl * WAVE,
                            pb.rowids.begin() + (size_t)(sl + 1) * WAVE);
          for (uint32_t q0 = 0; q0 < sm.n_quad; q0 += (uint32_t)batch) {
            BatchDesc d;
            d.quad0 = sm.first_quad + q0;
            d.row_slot = new_sl * WAVE;
            const uint32_t left = sm.n_quad - q0;
            d.info = left < (uint32_t)batch ? left : (uint32_t)batch;
            if (left <= (uint32_t)batch) d.info |= BD_LAST;
            if (first) d.info |= BD_TILE_FIRST;
            d.pad = 0;
            first = false;
            pb.descs.push_back(d);
          }
        }
        if (first) {  // no slice of this tile for this wave: barrier marker
          BatchDesc d;
          d.quad0 = 0;
          d.row_slot = 0;
          d.info = BD_TILE_FIRST;
          d.pad = 0;
          pb.descs.push_back(d);
        }
      }
      BatchDesc endd;
      endd.quad0 = 0;
      endd.row_slot = 0;
      endd.info = BD_END;
      endd.pad = 0;
      pb.descs.push_back(endd);
    }
  }
  // TileDesc::slice_begin/end keep describing the sorted order (set-up only;
  // the kernel reads col_block and the schedules).
  pb.ids.swap(new_ids);
  pb.vals.swap(new_vals);
  pb.rowids.swap(new_rowids);
  pb.slices.swap(new_slices);
}

struct VRow {
  int32_t begin;  // first entry (index into colidx)
  int32_t len;
  uint16_t slot;  // accumulator slot in LDS (row, or extra slot of a chunk)
  int32_t g_begin = 0;  // packed layout: first group in the tile's group list
  int32_t steps = 0;    // steps this row needs (sort key)
};

// Groups of one (chunk of a) row in the packed layout; see packed_row().
static int pack_groups(const int32_t* colidx, int32_t begin, int32_t len,
                       int64_t col0, std::vector<uint64_t>& out) {
  int n = 0;
  int32_t i = 0;
  while (i < len) {
    int64_t prev = colidx[begin + i] - col0;
    uint64_t g = (uint64_t)prev;
    int k = 1;
    while (k < 5 && i + k < len) {
      const int64_t d = (colidx[begin + i + k] - col0) - prev;
      if (d <= 0 || d > 4095) break;  // duplicate or long gap: new group
      g |= (uint64_t)d << (14 + 12 * (k - 1));
      prev += d;
      ++k;
    }
    out.push_back(g);
    i += k;
    ++n;
  }
  return n;
}

static void build_panel(int64_t R, int64_t C, const int32_t* rowptr,
                        const int32_t* colidx, const double* vals, int W,
                        int n_block, int PR, int G, int extra_budget,
                        int panel, bool packed, PanelBuild& pb) {
  const int64_t row0 = (int64_t)panel * PR;
  const int rows_here = (int)std::min<int64_t>(PR, R - row0);
  // pass 1: segment of every row in every column block
  std::vector<int32_t> seg_begin((size_t)rows_here * n_block),
      seg_len((size_t)rows_here * n_block);
  std::vector<int32_t> max_seg(rows_here, 0);
  for (int r = 0; r < rows_here; ++r) {
    int32_t k = rowptr[row0 + r];
    const int32_t e = rowptr[row0 + r + 1];
    for (int cb = 0; cb < n_block; ++cb) {
      const int64_t col_end = std::min<int64_t>((int64_t)(cb + 1) * W, C);
      const int32_t b = k;
      while (k < e && colidx[k] < col_end) ++k;
      seg_begin[(size_t)cb * rows_here + r] = b;
      seg_len[(size_t)cb * rows_here + r] = k - b;
      if (k - b > max_seg[r]) max_seg[r] = k - b;
    }
  }
  // split threshold T: the smallest one whose extra accumulators fit
  auto extras_for = [&](int T) {
    int64_t ex = 0;
    for (int r = 0; r < rows_here; ++r)
      if (max_seg[r] > T) ex += (max_seg[r] + T - 1) / T - 1;
    return ex;
  };
  int T = 0;  // 0 = no splitting
  {
    int longest = 0;
    for (int r = 0; r < rows_here; ++r) longest = std::max(longest, max_seg[r]);
    // never split below 3x the mean non-empty segment: balanced matrices
    // (e.g. the rows of X) gain nothing and would only get more slices
    int64_t seg_sum = 0, seg_cnt = 0;
    for (int32_t v : seg_len)
      if (v > 0) {
        seg_sum += v;
        ++seg_cnt;
      }
    int t_min = 32;
    if (seg_cnt > 0) {
      // Heavy-tailed segment lengths (the columns of simulate_data.py designs:
      // longest ~15x the mean) leave the waves that drew the long slices
      // streaming alone at the end of every tile; chunks of ~1.5x the mean
      // bring the busiest wave from 2.0x to 1.3x the ideal load (Tdot at
      // 1M x 50k: 60.8 -> 57.2 us).  Balanced rows (longest ~2.4x the mean)
      // only get more slices from splitting (dot: 54.3 -> 57.4 us), so they
      // keep the 3x rule.
      const double mean_seg = (double)seg_sum / (double)seg_cnt;
      double t_factor = (double)longest > 6. * mean_seg ? 1.5 : 3.;
      static const char* t_env = getenv("BBX_TILED_TFACTOR");
      if (t_env) t_factor = atof(t_env);
      t_min = std::max<int>(t_min, (int)(t_factor * mean_seg));
      // Small tiles: with fewer than ~1.5 slices per wave most of the 16
      // waves of the workgroup have nothing to stream (100k x 10k: 4 slices
      // per tile, busiest wave at 4-11x the ideal load).  There the split
      // threshold is lowered until every tile has ~2 slices per wave; the
      // chunks cost extra accumulators, which small panels have room for.
      int64_t densest_rows = 0, densest_entries = 0;
      for (int cb = 0; cb < n_block; ++cb) {
        int64_t rows_cb = 0, ent_cb = 0;
        for (int r = 0; r < rows_here; ++r) {
          const int32_t v = seg_len[(size_t)cb * rows_here + r];
          if (v > 0) {
            ++rows_cb;
            ent_cb += v;
          }
        }
        if (ent_cb > densest_entries) {
          densest_entries = ent_cb;
          densest_rows = rows_cb;
        }
      }
      const int64_t want_rows = 2 * TILE_WAVES * SLICE_ROWS;
      if (densest_rows < (3 * TILE_WAVES * SLICE_ROWS) / 2 && !t_env) {
        int t_par = (int)((densest_entries + want_rows - 1) / want_rows);
        t_par = (t_par + 3) / 4 * 4;
        if (t_par < 8) t_par = 8;
        if (t_par < t_min) t_min = t_par;
      }
    }
    if (extra_budget > 0 && longest > t_min) {
      int lo = t_min, hi = longest;  // extras_for(hi) == 0
      while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (extras_for(mid) <= extra_budget) hi = mid; else lo = mid + 1;
      }
      T = lo;
      if (T >= longest) T = 0;
    }
  }
  pb.split_T = T;
  // extra slots of the split rows
  std::vector<int32_t> extra_first(rows_here, -1);
  int n_extra = 0;
  if (T > 0)
    for (int r = 0; r < rows_here; ++r)
      if (max_seg[r] > T) {
        const int k = (max_seg[r] + T - 1) / T;
        extra_first[r] = PR + n_extra;
        FoldDesc fd;
        fd.row = (uint16_t)r;
        fd.first = (uint16_t)(PR + n_extra);
        fd.count = (uint16_t)(k - 1);
        fd.pad = 0;
        pb.folds.push_back(fd);
        n_extra += k - 1;
      }
  // pass 2: tiles
  pb.group_tile_count.assign(G, 0);
  const int blocks_per_group = (n_block + G - 1) / G;
  std::vector<VRow> vrows, sorted;
  std::vector<int> bucket;
  std::vector<uint64_t> groups;  // packed layout: groups of the tile's rows
  for (int cb = 0; cb < n_block; ++cb) {
    const int64_t col0 = (int64_t)cb * W;
    vrows.clear();
    int max_len = 0;
    for (int r = 0; r < rows_here; ++r) {
      const int32_t b = seg_begin[(size_t)cb * rows_here + r];
      const int32_t len = seg_len[(size_t)cb * rows_here + r];
      if (len == 0) continue;
      if (T > 0 && len > T) {
        const int k = (len + T - 1) / T;
        const int base = len / k, rem = len % k;
        int32_t at = b;
        for (int c = 0; c < k; ++c) {
          VRow v;
          v.begin = at;
          v.len = base + (c < rem ? 1 : 0);
          v.slot = (uint16_t)(c == 0 ? r : extra_first[r] + c - 1);
          at += v.len;
          vrows.push_back(v);
          max_len = std::max(max_len, v.len);
        }
      } else {
        VRow v;
        v.begin = b;
        v.len = len;
        v.slot = (uint16_t)r;
        vrows.push_back(v);
        max_len = std::max(max_len, len);
      }
    }
    // sort key: steps the row needs (4 entries per step, or its packed groups)
    groups.clear();
    int max_key = 0;
    for (VRow& v : vrows) {
      if (packed) {
        v.g_begin = (int32_t)groups.size();
        v.steps = pack_groups(colidx, v.begin, v.len, col0, groups);
      } else {
        v.steps = (v.len + 3) / 4;
      }
      max_key = std::max(max_key, v.steps);
    }
    // by decreasing step count (counting sort, stable)
    const int n_rows = (int)vrows.size();
    bucket.assign((size_t)max_key + 2, 0);
    for (const VRow& v : vrows) bucket[max_key - v.steps + 1] += 1;
    for (int b = 1; b <= max_key + 1; ++b) bucket[b] += bucket[b - 1];
    sorted.resize(vrows.size());
    for (const VRow& v : vrows) sorted[bucket[max_key - v.steps]++] = v;
    TileDesc td;
    td.col_block = cb;
    td.slice_begin = (int32_t)pb.slices.size();
    td.pad = 0;
    for (int base = 0; base < n_rows; base += SLICE_ROWS) {
      const int rows_in = std::min(SLICE_ROWS, n_rows - base);
      const uint32_t nq = (uint32_t)sorted[base].steps;  // longest row
      SliceMeta sm;
      sm.first_quad = (uint32_t)(pb.ids.size() / WAVE);
      sm.n_quad = nq;
      pb.slices.push_back(sm);
      const size_t id0 = pb.ids.size();
      pb.ids.resize(id0 + (size_t)nq * WAVE);
      if (vals) pb.vals.resize((id0 + (size_t)nq * WAVE) * 8, 0.);
      for (int l = 0; l < WAVE; ++l) {
        // lane l owns sorted rows base + l (A) and base + 64 + l (B)
        const VRow* vr[2] = {nullptr, nullptr};
        if (l < rows_in) vr[0] = &sorted[base + l];
        if (WAVE + l < rows_in) vr[1] = &sorted[base + WAVE + l];
        pb.rowids.push_back((uint32_t)(vr[0] ? vr[0]->slot : NO_ROW) |
                            ((uint32_t)(vr[1] ? vr[1]->slot : NO_ROW) << 16));
        for (uint32_t q = 0; q < nq && packed; ++q) {
          uint4 pk;
          uint64_t gg[2];
          for (int half = 0; half < 2; ++half) {
            const VRow* v = vr[half];
            gg[half] = (v && (int)q < v->steps) ? groups[(size_t)v->g_begin + q]
                                                : (uint64_t)W;  // xs[W] == 0
          }
          pk.x = (uint32_t)gg[0];
          pk.y = (uint32_t)(gg[0] >> 32);
          pk.z = (uint32_t)gg[1];
          pk.w = (uint32_t)(gg[1] >> 32);
          pb.ids[id0 + (size_t)q * WAVE + l] = pk;
        }
        for (uint32_t q = 0; q < nq && !packed; ++q) {
          uint16_t e[8];
          for (int half = 0; half < 2; ++half) {
            const VRow* v = vr[half];
            for (int u = 0; u < 4; ++u) {
              const int k = (int)q * 4 + u;
              if (v && k < v->len) {
                e[half * 4 + u] = (uint16_t)(colidx[v->begin + k] - col0);
                if (vals)
                  pb.vals[(id0 + (size_t)q * WAVE + l) * 8 + half * 4 + u] =
                      vals[v->begin + k];
              } else {
                e[half * 4 + u] = (uint16_t)W;  // xs[W] == 0
              }
            }
          }
          uint4 packed;
          packed.x = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
          packed.y = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
          packed.z = (uint32_t)e[4] | ((uint32_t)e[5] << 16);
          packed.w = (uint32_t)e[6] | ((uint32_t)e[7] << 16);
          pb.ids[id0 + (size_t)q * WAVE + l] = packed;
        }
      }
    }
    td.slice_end = (int32_t)pb.slices.size();
    pb.tiles.push_back(td);
    pb.group_tile_count[cb / blocks_per_group] += 1;
  }
  build_schedules(pb, G, vals ? BATCH_VAL : BATCH_BIN);
}

// Picks (PR, G): row panels x groups of column blocks.  One workgroup runs per
// CU (it owns the CU's LDS), so the launch should be a single round of <= 256
// workgroups of equal work.  Cost model fitted on MI355X (profiles/,
// DESIGN.md): a tile costs ~4.3 us of fixed time (slice refill from L2, two
// barriers, pipeline ramp) plus ~24 ps per stored entry streamed.
static void choose_shape(int64_t R, int64_t C, int64_t nnz, int n_block, int W,
                         int* PR_out, int* G_out) {
  double best = 1e300;
  int best_pr = 256, best_g = 1;
  const int lds_rows = (int)((TILE_LDS_BYTES - 2048) / 8) - (W + 8);
  int pr_cap = TILE_PR_MAX;
  if (lds_rows - 256 < pr_cap) pr_cap = lds_rows - 256;  // room for extras
  if (pr_cap < 128) pr_cap = 128;
  for (int pr = 128; pr <= pr_cap; pr += 128) {
    const int64_t n_panel = (R + pr - 1) / pr;
    for (int g = 1; g <= n_block; ++g) {
      const int bpg = (n_block + g - 1) / g;
      if ((n_block + bpg - 1) / bpg != g) continue;  // not a distinct split
      const double n_wg = (double)n_panel * g;
      const double rounds = std::ceil(n_wg / (double)TILE_WG_PER_ROUND);
      const double rows = (double)std::min<int64_t>(pr, R);
      const double tile_nnz = (double)nnz * rows / (double)R / n_block;
      const double per_tile = 4.3 + tile_nnz * 24e-6;            // us
      double cost = rounds * bpg * per_tile + 6.;
      if (g > 1) cost += (double)R * g * 16. / 4e6;              // slab pass
      if (cost < best) {
        best = cost;
        best_pr = pr;
        best_g = g;
      }
    }
  }
  *PR_out = best_pr;
  *G_out = best_g;
}

static int upload(DevMem& dst, const void* src, size_t bytes) {
  BBX_TRY(dst.alloc(bytes > 0 ? bytes : 8));
  if (bytes > 0) BBX_HIP(hipMemcpy(dst.ptr, src, bytes, hipMemcpyHostToDevice));
  return BBX_OK;
}

static int build_one(TiledMatrix& m, int64_t R, int64_t C, int64_t nnz,
                     const int32_t* rowptr, const int32_t* colidx,
                     const double* vals) {
  m.R = R;
  m.C = C;
  m.nnz = nnz;
  m.has_vals = vals != nullptr;
  {
    // Opt-in (BBX_TILED_PACK=1).  Measured at 1M x 50k: 18.5 % fewer id bytes
    // (233 -> 193 MB per product) but only 3-5 % less time (55.5 -> 53.9 us,
    // 57.4 -> 54.7 us): the per-entry work (LDS gather, index arithmetic) does
    // not shrink with the bytes, so the achieved HBM rate DROPS from 0.53 to
    // 0.45-0.48 of peak.  Kept for footprint-bound uses, not the default.
    static const char* pack_env = getenv("BBX_TILED_PACK");
    m.packed = !m.has_vals && pack_env && atoi(pack_env) == 1;
  }
  m.n_block = (int)((C + TILE_W_MAX - 1) / TILE_W_MAX);
  if (m.n_block < 1) m.n_block = 1;
  int64_t w = (C + m.n_block - 1) / m.n_block;
  w = (w + 63) / 64 * 64;
  m.W = (int)w;
  choose_shape(R, C, nnz, m.n_block, m.W, &m.PR, &m.G);
  if (const char* e = getenv("BBX_TILED_PR")) m.PR = atoi(e);
  if (const char* e = getenv("BBX_TILED_G")) m.G = atoi(e);
  if (m.PR < 64) m.PR = 64;
  if (m.PR > TILE_PR_MAX) m.PR = TILE_PR_MAX;
  if (m.G < 1) m.G = 1;
  if (m.G > m.n_block) m.G = m.n_block;
  {  // normalise G so that every group is non-empty
    const int bpg = (m.n_block + m.G - 1) / m.G;
    m.G = (m.n_block + bpg - 1) / bpg;
  }
  m.n_panel = (int)((R + m.PR - 1) / m.PR);
  // LDS left after the vector slice and the row accumulators pays for the
  // extra accumulators of split rows (2 KB stay free for static LDS).
  int extra_budget =
      (int)((TILE_LDS_BYTES - 2048) / 8) - (m.W + 8) - m.PR;
  if (extra_budget > 8192) extra_budget = 8192;
  if (extra_budget < 0) extra_budget = 0;
  if (const char* e = getenv("BBX_TILED_EXTRA")) extra_budget = atoi(e);

  std::vector<PanelBuild> pbs((size_t)m.n_panel);
  unsigned n_thr = std::thread::hardware_concurrency();
  if (n_thr < 1) n_thr = 1;
  if (n_thr > 64) n_thr = 64;
  if ((unsigned)m.n_panel < n_thr) n_thr = (unsigned)m.n_panel;
  std::vector<std::thread> pool;
  std::vector<int> thread_status(n_thr, BBX_OK);
  for (unsigned t = 0; t < n_thr; ++t)
    pool.emplace_back([&, t]() {
      // an exception must not leave a worker thread (std::terminate)
      thread_status[t] = no_throw([&]() -> int {
        for (int p = (int)t; p < m.n_panel; p += (int)n_thr)
          build_panel(R, C, rowptr, colidx, vals, m.W, m.n_block, m.PR, m.G,
                      extra_budget, p, m.packed, pbs[(size_t)p]);
        return BBX_OK;
      });
    });
  for (auto& th : pool) th.join();
  for (int st_t : thread_status)
    if (st_t < 0) return fail(BBX_ERR_INVALID, "out of host memory while tiling");

  if (getenv("BBX_TILED_STATS")) {
    int64_t crit_max = 0, total = 0, dup = 0, quads = 0, n_wg = 0, crit_sum = 0;
    int stat_extra = 0, stat_T = 0;
    for (auto& pb : pbs) {
      for (size_t g = 0; g < pb.wg_critical.size(); ++g) {
        crit_max = std::max(crit_max, pb.wg_critical[g]);
        crit_sum += pb.wg_critical[g];
        total += pb.wg_total[g];
        ++n_wg;
      }
      dup += pb.dup_quads;
      quads += (int64_t)(pb.ids.size() / WAVE);
      int ex = 0;
      for (const FoldDesc& fd : pb.folds) ex += fd.count;
      stat_extra = std::max(stat_extra, ex);
      if (pb.split_T > 0 && (stat_T == 0 || pb.split_T < stat_T))
        stat_T = pb.split_T;
    }
    fprintf(stderr,
            "[bbx tiled %lldx%lld] W=%d blocks=%d PR=%d G=%d split T=%d "
            "extras=%d workgroups=%lld: "
            "quads=%lld (+%lld re-loaded to fill batches, %.1f%%); batches per "
            "wave: ideal %.1f, mean critical path %.1f, worst workgroup %lld "
            "(%.1f%% over ideal)\n",
            (long long)R, (long long)C, m.W, m.n_block, m.PR, m.G, stat_T,
            stat_extra, (long long)n_wg, (long long)quads, (long long)dup,
            100. * (double)dup / (double)std::max<int64_t>(quads, 1),
            (double)total / (double)(n_wg * TILE_WAVES),
            (double)crit_sum / (double)n_wg, (long long)crit_max,
            100. * ((double)crit_max * n_wg * TILE_WAVES / (double)total - 1.));
  }
  // concatenate with offset fix-ups
  size_t tot_ids = 0, tot_slices = 0, tot_tiles = 0, tot_descs = 0;
  for (auto& pb : pbs) {
    tot_ids += pb.ids.size();
    tot_slices += pb.slices.size();
    tot_tiles += pb.tiles.size();
    tot_descs += pb.descs.size();
  }
  if (tot_ids / WAVE >= ((size_t)1 << 32))
    return fail(BBX_ERR_INVALID, "matrix too large for the tiled format");
  std::vector<uint4> ids(tot_ids);
  std::vector<double> vv(m.has_vals ? tot_ids * 8 : 0);
  std::vector<BatchDesc> descs(tot_descs);
  std::vector<int32_t> wave_desc((size_t)m.n_panel * m.G * TILE_WAVES, 0);
  std::vector<uint32_t> rowids(tot_slices * WAVE);
  std::vector<FoldDesc> folds;
  std::vector<int32_t> panel_fold((size_t)m.n_panel + 1, 0);
  m.n_extra = 0;
  m.split_T = 0;
  size_t id_off = 0, sl_off = 0, de_off = 0;
  for (int p = 0; p < m.n_panel; ++p) {
    PanelBuild& pb = pbs[(size_t)p];
    panel_fold[(size_t)p] = (int32_t)folds.size();
    int extra_here = 0;
    for (const FoldDesc& fd : pb.folds) {
      folds.push_back(fd);
      extra_here += fd.count;
    }
    if (extra_here > m.n_extra) m.n_extra = extra_here;
    if (pb.split_T > 0 && (m.split_T == 0 || pb.split_T < m.split_T))
      m.split_T = pb.split_T;
    if (!pb.ids.empty())
      memcpy(&ids[id_off], pb.ids.data(), pb.ids.size() * sizeof(uint4));
    if (m.has_vals && !pb.vals.empty())
      memcpy(&vv[id_off * 8], pb.vals.data(), pb.vals.size() * sizeof(double));
    for (size_t k = 0; k < pb.descs.size(); ++k) {
      BatchDesc d = pb.descs[k];
      if (d.info & 15u) {
        d.quad0 += (uint32_t)(id_off / WAVE);
        d.row_slot += (uint32_t)(sl_off * WAVE);
      }
      descs[de_off + k] = d;
    }
    for (size_t k = 0; k < pb.wave_desc.size(); ++k)
      wave_desc[(size_t)p * m.G * TILE_WAVES + k] =
          pb.wave_desc[k] + (int32_t)de_off;
    if (!pb.rowids.empty())
      memcpy(&rowids[sl_off * WAVE], pb.rowids.data(),
             pb.rowids.size() * sizeof(uint32_t));
    id_off += pb.ids.size();
    sl_off += pb.slices.size();
    de_off += pb.descs.size();
    std::vector<uint4>().swap(pb.ids);
    std::vector<double>().swap(pb.vals);
  }
  panel_fold[(size_t)m.n_panel] = (int32_t)folds.size();
  BBX_TRY(upload(m.folds, folds.data(), folds.size() * sizeof(FoldDesc)));
  BBX_TRY(upload(m.panel_fold, panel_fold.data(),
                 panel_fold.size() * sizeof(int32_t)));
  m.n_quad = (int64_t)(tot_ids / WAVE);
  m.n_slice = (int64_t)tot_slices;
  m.n_tile = (int64_t)tot_tiles;
  BBX_TRY(upload(m.ids, ids.data(), ids.size() * sizeof(uint4)));
  if (m.has_vals)
    BBX_TRY(upload(m.vals, vv.data(), vv.size() * sizeof(double)));
  m.n_desc = (int64_t)tot_descs;
  // the kernel addresses the streams with 32-bit byte offsets
  if ((uint64_t)tot_ids * (m.has_vals ? 64u : 16u) >= ((uint64_t)1 << 32))
    return fail(BBX_ERR_INVALID, "matrix too large for the tiled format");
  if (tot_slices * WAVE >= ((size_t)1 << 31) || tot_descs >= ((size_t)1 << 31))
    return fail(BBX_ERR_INVALID, "matrix too large for the tiled format");
  {  // equal-stride schedules when the padding stays small
    const size_t n_wave = wave_desc.size();
    size_t max_len = 0;
    for (size_t k = 0; k < n_wave; ++k) {
      const size_t end = (k + 1 < n_wave) ? (size_t)wave_desc[k + 1] : tot_descs;
      max_len = std::max(max_len, end - (size_t)wave_desc[k]);
    }
    const size_t stride = (max_len + WAVE - 1) / WAVE * WAVE;
    m.desc_stride = 0;
    if (n_wave > 0 && stride > 0 && n_wave * stride <= 2 * tot_descs + 65536 &&
        n_wave * stride < ((size_t)1 << 31)) {
      BatchDesc endd;
      endd.quad0 = 0;
      endd.row_slot = 0;
      endd.info = BD_END;
      endd.pad = 0;
      std::vector<BatchDesc> padded(n_wave * stride, endd);
      for (size_t k = 0; k < n_wave; ++k) {
        const size_t b = (size_t)wave_desc[k];
        const size_t end = (k + 1 < n_wave) ? (size_t)wave_desc[k + 1] : tot_descs;
        std::copy(descs.begin() + b, descs.begin() + end,
                  padded.begin() + k * stride);
      }
      descs.swap(padded);
      m.desc_stride = (int)stride;
    }
  }
  {  // the kernel prefetches descriptors in blocks of 64: keep reads in bounds
    BatchDesc endd;
    endd.quad0 = 0;
    endd.row_slot = 0;
    endd.info = BD_END;
    endd.pad = 0;
    descs.resize(descs.size() + 2 * WAVE, endd);
  }
  BBX_TRY(upload(m.descs, descs.data(), descs.size() * sizeof(BatchDesc)));
  BBX_TRY(upload(m.wave_desc, wave_desc.data(),
                 wave_desc.size() * sizeof(int32_t)));
  BBX_TRY(upload(m.rowids, rowids.data(), rowids.size() * sizeof(uint32_t)));
  BBX_TRY(m.slab.alloc(sizeof(double) * (size_t)m.G * (size_t)R));
  return BBX_OK;
}

static size_t lds_bytes(const TiledMatrix& m) {
  return sizeof(double) * ((size_t)m.W + 8 + (size_t)m.PR + (size_t)m.n_extra);
}

void destroy_tiled(bbx_design* h) {
  delete static_cast<TiledPair*>(h->tiled);
  h->tiled = nullptr;
}

// Builds both orientations from the device CSR arrays already in the handle
// (CSR of X and CSR of X^T) through a host pass.
int build_tiled(bbx_design* h) {
  const int64_t n = h->n, p = h->p, nnz = h->nnz;
  std::vector<int32_t> rowptr((size_t)n + 1), colidx((size_t)std::max<int64_t>(nnz, 1));
  std::vector<double> vals;
  BBX_HIP(hipMemcpy(rowptr.data(), h->indptr.ptr, sizeof(int32_t) * (size_t)(n + 1),
                    hipMemcpyDeviceToHost));
  if (nnz > 0)
    BBX_HIP(hipMemcpy(colidx.data(), h->indices.ptr, sizeof(int32_t) * (size_t)nnz,
                      hipMemcpyDeviceToHost));
  if (!h->binary) {
    vals.resize((size_t)std::max<int64_t>(nnz, 1));
    BBX_HIP(hipMemcpy(vals.data(), h->data.ptr, sizeof(double) * (size_t)nnz,
                      hipMemcpyDeviceToHost));
  }
  TiledPair* tp = new (std::nothrow) TiledPair();
  if (!tp) return fail(BBX_ERR_INVALID, "out of host memory");
  h->tiled = tp;
  BBX_TRY(build_one(tp->x, n, p, nnz, rowptr.data(), colidx.data(),
                    h->binary ? nullptr : vals.data()));
  // transpose orientation from the CSR of X^T built on the device
  rowptr.assign((size_t)p + 1, 0);
  BBX_HIP(hipMemcpy(rowptr.data(), h->t_indptr.ptr, sizeof(int32_t) * (size_t)(p + 1),
                    hipMemcpyDeviceToHost));
  if (nnz > 0)
    BBX_HIP(hipMemcpy(colidx.data(), h->t_indices.ptr,
                      sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost));
  if (!h->binary)
    BBX_HIP(hipMemcpy(vals.data(), h->t_data.ptr, sizeof(double) * (size_t)nnz,
                      hipMemcpyDeviceToHost));
  BBX_TRY(build_one(tp->xt, p, n, nnz, rowptr.data(), colidx.data(),
                    h->binary ? nullptr : vals.data()));
  for (const TiledMatrix* m : {&tp->x, &tp->xt}) {
    const size_t lb = lds_bytes(*m);
    if (lb > (size_t)TILE_LDS_BYTES)
      return fail(BBX_ERR_INVALID, "tile does not fit in LDS");
  }
#define BBX_TILED_ATTR(VV, PP, WW)                                             \
  BBX_HIP(hipFuncSetAttribute(                                                 \
      reinterpret_cast<const void*>(&tiled_spmv_kernel<VV, PP, WW>),           \
      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
  BBX_TILED_ATTR(false, false, false);
  BBX_TILED_ATTR(false, false, true);
  BBX_TILED_ATTR(false, true, false);
  BBX_TILED_ATTR(false, true, true);
  BBX_TILED_ATTR(true, false, false);
  BBX_TILED_ATTR(true, false, true);
#undef BBX_TILED_ATTR
  // The reference-layout arrays are only needed to build; free the big ones.
  if (!getenv("BBX_KEEP_CSR")) {
    h->indices.release();
    h->data.release();
    h->t_indices.release();
    h->t_data.release();
  }
  return BBX_OK;
}

static int launch_tiled(bbx_design* h, const TiledMatrix& m, const double* x,
                        const double* c_part, const double* x0_ptr,
                        const double* rowscale, double* out, double* slab,
                        double* out_sum_part, hipEvent_t ev_begin = nullptr,
                        hipEvent_t ev_end = nullptr) {
  const unsigned grid = (unsigned)(m.n_panel * m.G);
  const size_t lb = lds_bytes(m);
  static const int ablate = getenv("BBX_ABLATE") ? atoi(getenv("BBX_ABLATE")) : 0;
  // BBX_TILED_DEBUG=N: per-wave phase timing of launches N and N+1 (stderr)
  static const int dbg_at =
      getenv("BBX_TILED_DEBUG") ? atoi(getenv("BBX_TILED_DEBUG")) : -1;
  static int dbg_count = 0;
  static unsigned long long* dbg_buf = nullptr;
  unsigned long long* dbg = nullptr;
  if (dbg_at >= 0 && !BBX_TILED_INSTRUMENT)
    fprintf(stderr, "[bbx] BBX_TILED_DEBUG needs a -DBBX_TILED_INSTRUMENT=1 build\n");
  if (dbg_at >= 0 && BBX_TILED_INSTRUMENT) {
    if (!dbg_buf)
      BBX_HIP(hipMalloc(&dbg_buf, sizeof(unsigned long long) * 4096 * TILE_WAVES * 4));
    if ((dbg_count == dbg_at || dbg_count == dbg_at + 1) && grid <= 4096)
      dbg = dbg_buf;
    ++dbg_count;
  }
#define BBX_TILED_LAUNCH_W(VV, PP, WW, VALPTR)                                 \
  hipExtLaunchKernelGGL((tiled_spmv_kernel<VV, PP, WW>), dim3(grid),           \
                     dim3(TILE_THREADS), (unsigned)lb, h->stream, ev_begin,    \
                     ev_end, 0u, m.R, m.C, m.W, m.PR,                          \
                     m.G, (m.n_block + m.G - 1) / m.G,                         \
                     m.wave_desc.as<int32_t>(), m.desc_stride,                 \
                     m.descs.as<BatchDesc>(), m.rowids.as<uint32_t>(),         \
                     m.ids.as<uint4>(), VALPTR, x, c_part, x0_ptr, rowscale,   \
                     out, slab, m.PR + m.n_extra,                              \
                     m.panel_fold.as<int32_t>(), m.folds.as<FoldDesc>(),       \
                     out_sum_part, ablate, dbg)
#define BBX_TILED_LAUNCH(VV, PP, VALPTR)                                       \
  do {                                                                         \
    if (wide) BBX_TILED_LAUNCH_W(VV, PP, true, VALPTR);                        \
    else BBX_TILED_LAUNCH_W(VV, PP, false, VALPTR);                            \
  } while (0)
  // 16-byte slice loads need every slice start 16-byte aligned: W is a multiple
  // of 64 doubles, so it is the alignment of x itself that decides (inside the
  // CG loop x is an internal buffer placed accordingly; a caller's v + 1 of a
  // design with intercept is not, and takes the 8-byte path)
  static const bool no_wide =
      getenv("BBX_TILED_NARROW_FILL") && atoi(getenv("BBX_TILED_NARROW_FILL")) == 1;
  const bool wide = !no_wide && (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  if (m.has_vals)
    BBX_TILED_LAUNCH(true, false, m.vals.as<double>());
  else if (m.packed)
    BBX_TILED_LAUNCH(false, true, nullptr);
  else
    BBX_TILED_LAUNCH(false, false, nullptr);
#undef BBX_TILED_LAUNCH
#undef BBX_TILED_LAUNCH_W
  BBX_HIP(hipGetLastError());
  if (dbg) {
    BBX_HIP(hipStreamSynchronize(h->stream));
    const size_t nw = (size_t)grid * TILE_WAVES;
    std::vector<unsigned long long> hb(nw * 4);
    BBX_HIP(hipMemcpy(hb.data(), dbg, hb.size() * 8, hipMemcpyDeviceToHost));
    double tot = 0., loop = 0., sw = 0., skew = 0.;
    for (size_t w = 0; w < nw; ++w) {
      tot += (double)hb[w * 4];
      loop += (double)hb[w * 4 + 1];
      sw += (double)hb[w * 4 + 2];
      skew += (double)hb[w * 4 + 3];
    }
    fprintf(stderr,
            "[bbx tiled dbg grid=%u launch=%d] ticks per wave (mean): total "
            "%.0f = streaming %.0f + tile switches %.0f (waiting for the last "
            "wave %.0f, refill + second barrier %.0f) + epilogue %.0f\n",
            grid, dbg_count, tot / nw, (loop - sw) / nw, sw / nw, skew / nw,
            (sw - skew) / nw, (tot - loop) / nw);
  }
  return BBX_OK;
}

// Returns 1 through *sum_done when the partial sums of the output were
// produced by the kernel itself (no separate reduction pass needed).
int launch_dot_tiled(bbx_design* h, const double* d_v,
                     const double* d_rowscale, double* d_t,
                     double* d_sum_part, int* sum_done) {
  TiledPair* tp = static_cast<TiledPair*>(h->tiled);
  const TiledMatrix& m = tp->x;
  const double* x = d_v + h->intercept;
  const double* x0 = h->intercept ? d_v : nullptr;
  if (sum_done) *sum_done = 0;
  if (m.G == 1) {
    double* fused = nullptr;
    if (d_sum_part && m.n_panel <= NPART) {
      fused = d_sum_part;
      if (sum_done) *sum_done = 1;
    }
    hipEvent_t ea, eb;
    BBX_TRY(timer_arm(h, 0, &ea, &eb));
    return launch_tiled(h, m, x, part_slot(h, PS_C), x0, d_rowscale, d_t,
                        nullptr, fused, ea, eb);
  }
  // G > 1: two kernels in the family, bracketed by a pair of record commands
  BBX_TRY(timer_begin(h, 0));
  BBX_TRY(launch_tiled(h, m, x, nullptr, nullptr, nullptr, nullptr,
                       m.slab.as<double>(), nullptr));
  hipLaunchKernelGGL(tiled_dot_finalize_kernel, dim3(1024), dim3(256), 0,
                     h->stream, m.R, m.G, m.slab.as<double>(),
                     part_slot(h, PS_C), x0, d_rowscale, d_t);
  BBX_HIP(hipGetLastError());
  BBX_TRY(timer_end(h, 0));
  return BBX_OK;
}

int launch_tdot_tiled(bbx_design* h, const double* d_w,
                      const double* d_sumw_part, const TdotEpilogue& ep,
                      double* d_out) {
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

int launch_tdot_main_tiled(bbx_design* h, const double* d_w,
                           TdotSource* src) {
  TiledPair* tp = static_cast<TiledPair*>(h->tiled);
  const TiledMatrix& m = tp->xt;
  hipEvent_t ea, eb;
  BBX_TRY(timer_arm(h, 1, &ea, &eb));
  BBX_TRY(launch_tiled(h, m, d_w, nullptr, nullptr, nullptr, nullptr,
                       m.slab.as<double>(), nullptr, ea, eb));
  src->gfull = m.slab.as<double>();
  src->n_slab = m.G;
  src->stride = h->p;
  src->offset = h->offset.as<double>();
  src->p_eff = h->p;
  src->intercept = h->intercept;
  return BBX_OK;
}

int tiled_matvec_bytes(const bbx_design* h, int64_t* dot_bytes,
                       int64_t* tdot_bytes, bool timed_only) {
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
  if (timed_only)
    *tdot_bytes = tp->xt.stream_bytes() + 8 * h->n +
                  8 * (int64_t)tp->xt.G * h->p;
  else
    *tdot_bytes = tp->xt.stream_bytes() + 8 * (h->n + h->P) +
                  16 * (int64_t)tp->xt.G * h->p;
  return BBX_OK;
}

int64_t tiled_storage_bytes(const bbx_design* h) {
  const TiledPair* tp = static_cast<const TiledPair*>(h->tiled);
  if (!tp) return 0;
  return tp->x.stream_bytes() + tp->xt.stream_bytes();
}

int tiled_describe(const bbx_design* h, int which, int* W, int* n_block,
                   int* PR, int* G, int64_t* n_quad, int64_t* n_slice) {
  const TiledPair* tp = static_cast<const TiledPair*>(h->tiled);
  if (!tp) return fail(BBX_ERR_STATE, "tiled format not built");
  const TiledMatrix& m = which == 0 ? tp->x : tp->xt;
  if (W) *W = m.W;
  if (n_block) *n_block = m.n_block;
  if (PR) *PR = m.PR;
  if (G) *G = m.G;
  if (n_quad) *n_quad = m.n_quad;
  if (n_slice) *n_slice = m.n_slice;
  return BBX_OK;
}

}  // namespace bbx
