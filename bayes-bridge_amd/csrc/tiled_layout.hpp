// Host-side description of the LDS-tiled sparse layout (BBX_FORMAT_TILED):
// the structs the kernel reads, the builder that produces them from a CSR
// matrix, and a CPU emulator of the kernel's walk.  Plain C++17, no HIP: this
// header and tiled_layout.cpp also build with g++ (sanitizers, CPU tests).
// The layout itself is described at the top of spmv_tiled.hip and in DESIGN.md.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace bbx {

// Workgroup geometry.  Default: one 1024-thread workgroup owns a CU's LDS.
// -DBBX_TILE_THREADS=512 -DBBX_TILE_LDS_KB=80 -DBBX_TILE_W_MAX=7168 builds the
// "two half-width workgroups per CU" variant measured in LABNOTES.md 3.1.
#ifndef BBX_TILE_THREADS
#define BBX_TILE_THREADS 1024
#endif
#ifndef BBX_TILE_LDS_KB
#define BBX_TILE_LDS_KB 160
#endif
#ifndef BBX_TILE_W_MAX
#define BBX_TILE_W_MAX 16128
#endif
#ifndef BBX_BATCH_BIN
#define BBX_BATCH_BIN 1
#endif
constexpr int LANES = 64;                   // gfx950 wavefront
constexpr int TILE_W_MAX = BBX_TILE_W_MAX;  // doubles of the vector slice in LDS
constexpr int TILE_PR_MAX = 8192;           // row accumulators in LDS
constexpr int TILE_THREADS = BBX_TILE_THREADS;
constexpr int TILE_LDS_BYTES = BBX_TILE_LDS_KB * 1024;
constexpr int TILE_WG_PER_CU = (160 / BBX_TILE_LDS_KB);
constexpr int TILE_WG_PER_ROUND = 256 * TILE_WG_PER_CU;
constexpr int TILE_WAVES = TILE_THREADS / LANES;
constexpr uint16_t NO_ROW = 0xFFFF;
constexpr int SLICE_ROWS = 2 * LANES;  // two rows per lane
constexpr int BATCH_BIN = BBX_BATCH_BIN;  // steps per ring slot, value-free
constexpr int BATCH_VAL = 1;   // steps per ring slot when values are stored

// 16 bytes of ids: one step of one lane (same layout as HIP's uint4).
struct Ids4 {
  uint32_t x, y, z, w;
};
static_assert(sizeof(Ids4) == 16, "one 16-byte lane load");

// Packed value-free steps (one right-hand side; TiledHost::packed).  The 16
// bytes of a lane and step hold one GROUP per row -- x | y << 32 for row A,
// z | w << 32 for row B -- instead of four 16-bit ids per row: bits 0-13 the
// slice slot of the group's first entry, then four 12-bit forward deltas, five
// entries in eight bytes.  The kernel always gathers all five slots
// (s_{k+1} = s_k + d_k): there is no "absent" marker to test.  Instead the
// slice in LDS carries ZERO SLOTS -- one after every 4095 columns (slots 4095,
// 8191, ...) and one behind the last column -- so that from every column's slot
// a zero slot lies at most 4095 slots ahead: a group with fewer than five
// entries steps onto the next zero slot and stays there (delta 0), adding 0.0.
// A delta of 0 between entries is a duplicate column (counted twice, as CSR
// semantics demand); entries more than 4095 slots apart start a new group.
constexpr int PACK_PERIOD = 4095;  // columns between two zero slots
inline int packed_slot(int j) { return j + j / PACK_PERIOD; }   // column -> slot
inline int packed_slots(int W) { return W + (W - 1) / PACK_PERIOD; }  // = terminal zero slot

// Set-up only (the kernel derives the column block arithmetically).
struct TileDesc {
  int32_t col_block;
  int32_t slice_begin;
  int32_t slice_end;
  int32_t pad;
};

struct SliceMeta {
  uint32_t first_quad;  // offset into the id stream in units of 64 x 16 bytes
  uint32_t n_quad;      // steps: 4 entries (packed: one group) of row A + of row B per lane
};

// One step of a wave's precomputed schedule: BATCH consecutive quads of one
// slice.  The schedule of every (workgroup, wave) is laid out in processing
// order, so the kernel's issue cursor is a single scalar index.
struct BatchDesc {
  uint32_t quad0;     // first step (units of 64 x 16 bytes), counted from the
                      // first step of the WORKGROUP's stretch of the stream
                      // (TiledHost::wg_quad0): the kernel forms 64-bit bases per
                      // workgroup and 32-bit byte offsets inside them, so a
                      // stream may exceed 4 GiB
  uint32_t row_slot;  // slice * 64: where the slice's row-id pairs start
  uint32_t info;      // bits 0-3 count, 8 last-of-slice, 9 tile-first, 10 end
  uint32_t pad;
};
constexpr uint32_t BD_LAST = 1u << 8;
constexpr uint32_t BD_TILE_FIRST = 1u << 9;
constexpr uint32_t BD_END = 1u << 10;

// Row r of a panel was split: acc[r] += acc[first .. first+count) at the end.
struct FoldDesc {
  uint16_t row, first, count, pad;
};

// Build-time choices that the environment can override (diagnostics, A/B).
struct TiledOptions {
  // Right-hand sides that share one pass over the matrix (batched chains):
  // the LDS holds K interleaved vector slices and K accumulators per row, so W
  // and PR shrink to what K of each fit next to each other.  1, 2 or 4.
  int chains = 1;
  int force_PR = 0;          // BBX_TILED_PR
  int force_G = 0;           // BBX_TILED_G
  int force_blocks = 0;      // number of column blocks (0: automatic)
  int extra_budget = -1;     // extra accumulators per panel (< 0: what LDS leaves)
  double t_factor = 0.;      // split threshold / mean segment (0: automatic)
  bool bank_aware = true;    // bank-aware entry order inside the rows
  // value-free ids of a single right-hand side as groups of five (see
  // packed_slot): -1 = whichever form stores fewer steps, 0 / 1 = forced
  int packed = -1;
  bool stats = false;        // BBX_TILED_STATS=1
  int max_threads = 64;
  // transpose: the X^T orientation reads BBX_TILED_PR_T / BBX_TILED_G_T first
  static TiledOptions from_env(bool transpose = false);
};

// One orientation (X or X^T) in tiled form, host resident.
struct TiledHost {
  int64_t R = 0, C = 0, nnz = 0;
  int W = 0, n_block = 0, PR = 0, n_panel = 0, G = 0;
  int K = 1;  // right-hand sides the geometry was sized for (TiledOptions::chains)
  bool has_vals = false;
  bool packed = false;  // steps hold one 5-entry group per row (packed_slot)
  int Wl = 0;           // slots of one vector slice in LDS: W, or packed_slots(W)
  int64_t n_slice = 0, n_quad = 0, n_tile = 0, n_desc = 0;
  int desc_stride = 0;  // > 0: wave k's schedule starts at k * desc_stride
  int n_extra = 0;      // extra accumulators per panel (row splitting)
  int split_T = 0;      // smallest split threshold used by any panel (0 = none)
  std::vector<Ids4> ids;            // [n_quad * 64]
  std::vector<double> vals;         // [n_quad * 64 * 8] when has_vals
  std::vector<BatchDesc> descs;     // per-wave schedules (+ 2 blocks of END)
  std::vector<int32_t> wave_desc;   // [n_panel * G * TILE_WAVES]
  std::vector<uint32_t> wg_quad0;   // [n_panel * G] first step of each workgroup
  std::vector<uint32_t> rowids;     // [n_slice * 64]: panel-local rows A | B<<16
  std::vector<FoldDesc> folds;
  std::vector<int32_t> panel_fold;  // [n_panel + 1]
  std::string stats;                // filled when TiledOptions::stats
  double model_cost_us = 0.;        // the geometry search's estimate of one product
  int64_t lds_doubles() const {
    return (int64_t)K * ((int64_t)Wl + 8 + PR + n_extra);
  }
};

// Builds the tiled form of an R x C CSR matrix (`vals` == nullptr: every
// stored value is 1.0).  Returns 0, or -1 with *err set (too large for the
// format's 32-bit offsets, tile does not fit in LDS, out of memory).
int build_tiled_host(int64_t R, int64_t C, int64_t nnz, const int64_t* rowptr,
                     const int32_t* colidx, const double* vals,
                     const TiledOptions& opt, TiledHost* out, std::string* err);

// The geometry search's cost estimate (microseconds per product) of the layout
// build_tiled_host would choose for `chains` right-hand sides; no layout is
// built.  < 0 on bad arguments.
double tiled_model_cost(int64_t R, int64_t C, int64_t nnz,
                        const int64_t* rowptr, int chains);

// Worker threads of the builder: affinity mask, capped by the cgroup CPU quota,
// divided by LOCAL_WORLD_SIZE, capped by max_threads (BBX_BUILD_THREADS overrides).
int builder_threads(int max_threads);
// Host cores this rank can keep busy: affinity mask, capped by the cgroup CPU
// quota, divided by LOCAL_WORLD_SIZE (what builder_threads starts from).
int cores_per_rank();

// CPU emulation of tiled_spmv_kernel's walk over the layout: every workgroup,
// every wave's schedule, lane-private sums flushed into the panel's
// accumulators at the end of each slice, split rows folded in descriptor
// order.  slab[g * R + r] = partial sum of row r over the column blocks of
// group g -- the same additions in the same order as the kernel, so a GPU
// launch can be compared with it bit for bit.  (A layout sized for K > 1
// right-hand sides is walked one right-hand side at a time: the batched kernel
// does the same additions in the same order for each of its K columns.)
void emulate_tiled_spmv(const TiledHost& m, const double* x,
                        std::vector<double>* slab);

// A CSR matrix in host memory as the builder reads it: 64-bit row pointers,
// 32-bit column ids, values (empty: every stored value is 1.0).
struct HostCsr {
  std::vector<int64_t> rowptr;
  std::vector<int32_t> colidx;
  std::vector<double> vals;
};

// Structure check of a host CSR whose index arrays are 64-bit (what SciPy holds
// once a matrix has 2^31 or more stored entries, or was built from int64
// arrays): 0 = fine, else the bits validate_csr_kernel (api.hip) reports --
// 1 row pointers not 0 ... nnz non-decreasing, 2 column id out of range,
// 4 column ids of a row not ascending.
int check_csr64_host(int64_t R, int64_t C, int64_t nnz, const int64_t* rowptr,
                     const int64_t* colidx, int max_threads);

// X^T of an R x C host CSR by a stable counting sort over the column ids (the
// rows of X^T come out with ascending ids, duplicates in stored order: exactly
// what the device transposition, a stable radix sort, produces).  `vals` may be
// nullptr.  Throws std::bad_alloc.
void transpose_csr_host(int64_t R, int64_t C, const int64_t* rowptr,
                        const int32_t* colidx, const double* vals,
                        int max_threads, HostCsr* out);

// LDS bank statistics of the gathers: over every (slice, step, entry position,
// 32-lane half) the number of LDS cycles a ds_read_b64 needs (1 = conflict
// free).  Returns the mean.
double tiled_mean_gather_cycles(const TiledHost& m);

}  // namespace bbx
