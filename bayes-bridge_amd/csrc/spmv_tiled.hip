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
//     grouped 64 at a time into slices (one wavefront each, lane = row), stored
//     "sliced ELL": [quad q][lane][4] block-local uint16 column ids, padded to
//     a multiple of 4 entries with an id that points at a 0.0 in LDS;
//   * values are stored only when some entry differs from 1.0 (binary designs
//     of simulate_data.py:100-117 never read values; the unwired prototype
//     design_matrix/cython_matmal/binary_matmul.pyx:21-25 had the same idea).
//
// One workgroup (1024 threads, 16 waves, one per CU) owns a row panel and a
// group of column blocks: it fills the vector slice, streams the tile's ids
// with coalesced 512-byte wave loads (the only HBM traffic that scales with
// nnz: 2 bytes per entry), adds lane-private sums into LDS accumulators, and
// writes the panel once.  No atomics: every sum has a fixed order.
#include <algorithm>
#include <cstring>
#include <thread>

#include "common.hpp"

namespace bbx {

constexpr int TILE_W_MAX = 16128;  // doubles of the vector slice in LDS
constexpr int TILE_PR_MAX = 4096;  // row accumulators in LDS
constexpr int TILE_THREADS = 1024;
constexpr int TILE_WAVES = TILE_THREADS / WAVE;
constexpr uint16_t NO_ROW = 0xFFFF;

struct TileDesc {
  int32_t col_block;
  int32_t slice_begin;
  int32_t slice_end;
  int32_t pad;
};

struct SliceMeta {
  uint32_t first_quad;  // offset into the id stream in units of 64 uint2
  uint32_t n_quad;      // groups of 4 entries per lane
};

// One orientation (X or X^T) in tiled form, device resident.
struct TiledMatrix {
  int64_t R = 0, C = 0, nnz = 0;
  int W = 0, n_block = 0, PR = 0, n_panel = 0, G = 0;
  bool has_vals = false;
  int64_t n_slice = 0, n_quad = 0, n_tile = 0;
  DevMem ids;        // uint2[n_quad * 64]
  DevMem vals;       // double[n_quad * 64 * 4] when has_vals
  DevMem slices;     // SliceMeta[n_slice]
  DevMem rowids;     // uint16[n_slice * 64] panel-local row of each lane
  DevMem tiles;      // TileDesc[n_tile]
  DevMem wg_tiles;   // int32[n_panel * G + 1]
  DevMem slab;       // double[G * R] partial sums when G > 1 (or Tdot)
  int64_t stream_bytes() const {
    return (int64_t)n_quad * 64 * 8 * (has_vals ? 5 : 1) +
           (int64_t)n_slice * (sizeof(SliceMeta) + 128) +
           (int64_t)n_tile * (int64_t)sizeof(TileDesc);
  }
};

struct TiledPair {
  TiledMatrix x, xt;
};

// ------------------------------------------------------------------ kernel

template <bool VALS>
__global__ __launch_bounds__(TILE_THREADS) void tiled_spmv_kernel(
    int64_t R, int64_t C, int W, int PR, int G,
    const int32_t* __restrict__ wg_tiles, const TileDesc* __restrict__ tiles,
    const SliceMeta* __restrict__ slices, const uint16_t* __restrict__ rowids,
    const uint2* __restrict__ ids, const double* __restrict__ vals,
    const double* __restrict__ x,
    // epilogue (direct mode, G == 1 and out != nullptr):
    //   out[r] = rowscale[r] * (c0 - sum(c_part) + acc)
    const double* __restrict__ c_part, const double* x0_ptr,
    const double* __restrict__ rowscale, double* __restrict__ out,
    double* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* xs = lds;               // W + 8 doubles; xs[W] == 0 (padding target)
  double* acc = lds + (W + 8);    // PR doubles
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wave = tid / WAVE;
  const int panel = blockIdx.x / G;
  const int group = blockIdx.x - panel * G;
  const int64_t row0 = (int64_t)panel * PR;
  const int rows_here = (int)((R - row0 < PR) ? (R - row0) : PR);

  for (int r = tid; r < PR; r += TILE_THREADS) acc[r] = 0.;
  if (tid < 8) xs[W + tid] = 0.;

  const int t_begin = wg_tiles[blockIdx.x], t_end = wg_tiles[blockIdx.x + 1];
  for (int t = t_begin; t < t_end; ++t) {
    const TileDesc td = tiles[t];
    const int64_t col0 = (int64_t)td.col_block * W;
    const int cols_here = (int)((C - col0 < W) ? (C - col0) : W);
    __syncthreads();  // previous tile's gathers are done
    for (int j = tid; j < W; j += TILE_THREADS)
      xs[j] = (j < cols_here) ? x[col0 + j] : 0.;
    __syncthreads();
    for (int s = td.slice_begin + wave; s < td.slice_end; s += TILE_WAVES) {
      const SliceMeta sm = slices[s];
      const uint16_t rid = rowids[(int64_t)s * WAVE + lane];
      const uint2* __restrict__ p = ids + (int64_t)sm.first_quad * WAVE + lane;
      const double* __restrict__ pv =
          VALS ? vals + ((int64_t)sm.first_quad * WAVE + lane) * 4 : nullptr;
      double s0 = 0., s1 = 0.;
      uint32_t q = 0;
      for (; q + 4 <= sm.n_quad; q += 4) {
        const uint2 a = p[(q + 0) * WAVE];
        const uint2 b = p[(q + 1) * WAVE];
        const uint2 c = p[(q + 2) * WAVE];
        const uint2 d = p[(q + 3) * WAVE];
        if (VALS) {
          const double* v = pv + (int64_t)q * WAVE * 4;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint2 e = (u == 0) ? a : (u == 1) ? b : (u == 2) ? c : d;
            const double4 vv =
                *reinterpret_cast<const double4*>(v + (int64_t)u * WAVE * 4);
            s0 += vv.x * xs[e.x & 0xFFFFu] + vv.z * xs[e.y & 0xFFFFu];
            s1 += vv.y * xs[e.x >> 16] + vv.w * xs[e.y >> 16];
          }
        } else {
          s0 += xs[a.x & 0xFFFFu] + xs[a.y & 0xFFFFu];
          s1 += xs[a.x >> 16] + xs[a.y >> 16];
          s0 += xs[b.x & 0xFFFFu] + xs[b.y & 0xFFFFu];
          s1 += xs[b.x >> 16] + xs[b.y >> 16];
          s0 += xs[c.x & 0xFFFFu] + xs[c.y & 0xFFFFu];
          s1 += xs[c.x >> 16] + xs[c.y >> 16];
          s0 += xs[d.x & 0xFFFFu] + xs[d.y & 0xFFFFu];
          s1 += xs[d.x >> 16] + xs[d.y >> 16];
        }
      }
      for (; q < sm.n_quad; ++q) {
        const uint2 a = p[q * WAVE];
        if (VALS) {
          const double4 vv =
              *reinterpret_cast<const double4*>(pv + (int64_t)q * WAVE * 4);
          s0 += vv.x * xs[a.x & 0xFFFFu] + vv.z * xs[a.y & 0xFFFFu];
          s1 += vv.y * xs[a.x >> 16] + vv.w * xs[a.y >> 16];
        } else {
          s0 += xs[a.x & 0xFFFFu] + xs[a.y & 0xFFFFu];
          s1 += xs[a.x >> 16] + xs[a.y >> 16];
        }
      }
      if (rid != NO_ROW) acc[rid] += s0 + s1;
    }
  }
  __syncthreads();
  if (out) {
    // direct epilogue: c = x0 - sum(c_part), summed once in a fixed order
    if (tid < WAVE) {
      double cs = 0.;
      if (c_part) {
#pragma unroll
        for (int k = 0; k < NPART / WAVE; ++k) cs += c_part[tid + k * WAVE];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cs += __shfl_down(cs, off, WAVE);
      }
      if (tid == 0) xs[0] = (x0_ptr ? *x0_ptr : 0.) - cs;
    }
    __syncthreads();
    const double c = xs[0];
    for (int r = tid; r < rows_here; r += TILE_THREADS) {
      double v = c + acc[r];
      if (rowscale) v *= rowscale[row0 + r];
      out[row0 + r] = v;
    }
  } else {
    double* dst = slab + (int64_t)group * R + row0;
    for (int r = tid; r < rows_here; r += TILE_THREADS) dst[r] = acc[r];
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

// gfull[r] = sum_g slab[g][r]   (Tdot: feeds tdot_finalize_kernel)
__global__ __launch_bounds__(256) void tiled_slab_sum_kernel(
    int64_t R, int G, const double* __restrict__ slab,
    double* __restrict__ gfull) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < R;
       r += (int64_t)gridDim.x * 256) {
    double a = 0.;
    for (int g = 0; g < G; ++g) a += slab[(int64_t)g * R + r];
    gfull[r] = a;
  }
}

// ----------------------------------------------------------------- builder

struct PanelBuild {
  std::vector<uint2> ids;
  std::vector<double> vals;
  std::vector<SliceMeta> slices;  // first_quad local to the panel
  std::vector<uint16_t> rowids;
  std::vector<TileDesc> tiles;    // slice ids local to the panel
  std::vector<int32_t> group_tile_count;
};

static void build_panel(int64_t R, int64_t C, const int32_t* rowptr,
                        const int32_t* colidx, const double* vals, int W,
                        int n_block, int PR, int G, int panel,
                        PanelBuild& pb) {
  const int64_t row0 = (int64_t)panel * PR;
  const int rows_here = (int)std::min<int64_t>(PR, R - row0);
  std::vector<int32_t> cursor(rows_here), seg_begin(rows_here),
      seg_len(rows_here);
  for (int r = 0; r < rows_here; ++r) cursor[r] = rowptr[row0 + r];
  pb.group_tile_count.assign(G, 0);
  const int blocks_per_group = (n_block + G - 1) / G;
  std::vector<int> order;
  std::vector<int> bucket;
  for (int cb = 0; cb < n_block; ++cb) {
    const int64_t col_end = std::min<int64_t>((int64_t)(cb + 1) * W, C);
    const int64_t col0 = (int64_t)cb * W;
    int max_len = 0;
    for (int r = 0; r < rows_here; ++r) {
      const int32_t e = rowptr[row0 + r + 1];
      int32_t k = cursor[r];
      seg_begin[r] = k;
      while (k < e && colidx[k] < col_end) ++k;
      seg_len[r] = k - cursor[r];
      cursor[r] = k;
      if (seg_len[r] > max_len) max_len = seg_len[r];
    }
    // rows with entries, by decreasing count (counting sort, stable)
    bucket.assign((size_t)max_len + 2, 0);
    int n_rows = 0;
    for (int r = 0; r < rows_here; ++r)
      if (seg_len[r] > 0) {
        bucket[max_len - seg_len[r] + 1] += 1;
        ++n_rows;
      }
    for (int b = 1; b <= max_len + 1; ++b) bucket[b] += bucket[b - 1];
    order.assign(n_rows, 0);
    for (int r = 0; r < rows_here; ++r)
      if (seg_len[r] > 0) order[bucket[max_len - seg_len[r]]++] = r;
    TileDesc td;
    td.col_block = cb;
    td.slice_begin = (int32_t)pb.slices.size();
    td.pad = 0;
    for (int base = 0; base < n_rows; base += WAVE) {
      const int lanes = std::min(WAVE, n_rows - base);
      const int len = seg_len[order[base]];  // longest row of the slice
      const uint32_t nq = (uint32_t)((len + 3) / 4);
      SliceMeta sm;
      sm.first_quad = (uint32_t)(pb.ids.size() / WAVE);
      sm.n_quad = nq;
      pb.slices.push_back(sm);
      const size_t id0 = pb.ids.size();
      pb.ids.resize(id0 + (size_t)nq * WAVE);
      if (vals) pb.vals.resize((id0 + (size_t)nq * WAVE) * 4, 0.);
      for (int l = 0; l < WAVE; ++l) {
        if (l < lanes) {
          const int r = order[base + l];
          pb.rowids.push_back((uint16_t)r);
          const int32_t b = seg_begin[r];
          const int n_ent = seg_len[r];
          for (uint32_t q = 0; q < nq; ++q) {
            uint16_t e[4];
            for (int u = 0; u < 4; ++u) {
              const int k = (int)q * 4 + u;
              if (k < n_ent) {
                e[u] = (uint16_t)(colidx[b + k] - col0);
                if (vals)
                  pb.vals[((id0 + (size_t)q * WAVE + l) * 4) + u] =
                      vals[b + k];
              } else {
                e[u] = (uint16_t)W;  // xs[W] == 0
              }
            }
            uint2 packed;
            packed.x = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
            packed.y = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
            pb.ids[id0 + (size_t)q * WAVE + l] = packed;
          }
        } else {
          pb.rowids.push_back(NO_ROW);
          for (uint32_t q = 0; q < nq; ++q) {
            uint2 packed;
            packed.x = (uint32_t)W | ((uint32_t)W << 16);
            packed.y = packed.x;
            pb.ids[id0 + (size_t)q * WAVE + l] = packed;
          }
        }
      }
    }
    td.slice_end = (int32_t)pb.slices.size();
    pb.tiles.push_back(td);
    pb.group_tile_count[cb / blocks_per_group] += 1;
  }
}

// Picks (PR, G): panels x groups of column blocks ~ one or two waves of
// workgroups over the 256 CUs while keeping the LDS refills (W*8 bytes per
// tile, from L2) small next to the tile's id stream (2 bytes per entry).
static void choose_shape(int64_t R, int64_t C, int64_t nnz, int n_block, int W,
                         int* PR_out, int* G_out) {
  double best = 1e300;
  int best_pr = 256, best_g = 1;
  const int prs[] = {4096, 2048, 1024, 512, 256};
  for (int pr : prs) {
    if (pr > 256 && (int64_t)pr > R * 2) continue;
    const int64_t n_panel = (R + pr - 1) / pr;
    for (int g = 1; g <= n_block; ++g) {
      const int bpg = (n_block + g - 1) / g;
      if ((n_block + bpg - 1) / bpg != g) continue;  // not a distinct split
      const double n_wg = (double)n_panel * g;
      const double rounds = std::ceil(n_wg / 256.);
      const double tile_nnz = (double)nnz / ((double)n_panel * n_block);
      const double per_tile = W * 8. / 48. + tile_nnz * 2. / 8. + 400.;
      double cost = rounds * bpg * per_tile + 2000.;
      if (g > 1) cost += (double)R * g * 16. / (256. * 16.);
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

  std::vector<PanelBuild> pbs((size_t)m.n_panel);
  unsigned n_thr = std::thread::hardware_concurrency();
  if (n_thr < 1) n_thr = 1;
  if (n_thr > 64) n_thr = 64;
  if ((unsigned)m.n_panel < n_thr) n_thr = (unsigned)m.n_panel;
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < n_thr; ++t)
    pool.emplace_back([&, t]() {
      for (int p = (int)t; p < m.n_panel; p += (int)n_thr)
        build_panel(R, C, rowptr, colidx, vals, m.W, m.n_block, m.PR, m.G, p,
                    pbs[(size_t)p]);
    });
  for (auto& th : pool) th.join();

  // concatenate with offset fix-ups
  size_t tot_ids = 0, tot_slices = 0, tot_tiles = 0;
  for (auto& pb : pbs) {
    tot_ids += pb.ids.size();
    tot_slices += pb.slices.size();
    tot_tiles += pb.tiles.size();
  }
  if (tot_ids / WAVE >= ((size_t)1 << 32))
    return fail(BBX_ERR_INVALID, "matrix too large for the tiled format");
  std::vector<uint2> ids(tot_ids);
  std::vector<double> vv(m.has_vals ? tot_ids * 4 : 0);
  std::vector<SliceMeta> slices(tot_slices);
  std::vector<uint16_t> rowids(tot_slices * WAVE);
  std::vector<TileDesc> tiles(tot_tiles);
  std::vector<int32_t> wg_tiles((size_t)m.n_panel * m.G + 1, 0);
  size_t id_off = 0, sl_off = 0, ti_off = 0;
  for (int p = 0; p < m.n_panel; ++p) {
    PanelBuild& pb = pbs[(size_t)p];
    if (!pb.ids.empty())
      memcpy(&ids[id_off], pb.ids.data(), pb.ids.size() * sizeof(uint2));
    if (m.has_vals && !pb.vals.empty())
      memcpy(&vv[id_off * 4], pb.vals.data(), pb.vals.size() * sizeof(double));
    for (size_t s = 0; s < pb.slices.size(); ++s) {
      SliceMeta sm = pb.slices[s];
      sm.first_quad += (uint32_t)(id_off / WAVE);
      slices[sl_off + s] = sm;
    }
    if (!pb.rowids.empty())
      memcpy(&rowids[sl_off * WAVE], pb.rowids.data(),
             pb.rowids.size() * sizeof(uint16_t));
    for (size_t t = 0; t < pb.tiles.size(); ++t) {
      TileDesc td = pb.tiles[t];
      td.slice_begin += (int32_t)sl_off;
      td.slice_end += (int32_t)sl_off;
      tiles[ti_off + t] = td;
    }
    int32_t run = (int32_t)ti_off;
    for (int g = 0; g < m.G; ++g) {
      wg_tiles[(size_t)p * m.G + g] = run;
      run += pb.group_tile_count[(size_t)g];
    }
    id_off += pb.ids.size();
    sl_off += pb.slices.size();
    ti_off += pb.tiles.size();
    std::vector<uint2>().swap(pb.ids);
    std::vector<double>().swap(pb.vals);
  }
  wg_tiles[(size_t)m.n_panel * m.G] = (int32_t)ti_off;
  m.n_quad = (int64_t)(tot_ids / WAVE);
  m.n_slice = (int64_t)tot_slices;
  m.n_tile = (int64_t)tot_tiles;
  BBX_TRY(upload(m.ids, ids.data(), ids.size() * sizeof(uint2)));
  if (m.has_vals)
    BBX_TRY(upload(m.vals, vv.data(), vv.size() * sizeof(double)));
  BBX_TRY(upload(m.slices, slices.data(), slices.size() * sizeof(SliceMeta)));
  BBX_TRY(upload(m.rowids, rowids.data(), rowids.size() * sizeof(uint16_t)));
  BBX_TRY(upload(m.tiles, tiles.data(), tiles.size() * sizeof(TileDesc)));
  BBX_TRY(upload(m.wg_tiles, wg_tiles.data(),
                 wg_tiles.size() * sizeof(int32_t)));
  BBX_TRY(m.slab.alloc(sizeof(double) * (size_t)m.G * (size_t)R));
  return BBX_OK;
}

static size_t lds_bytes(const TiledMatrix& m) {
  return sizeof(double) * ((size_t)m.W + 8 + (size_t)m.PR);
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
  BBX_TRY(h->tiled_gfull.alloc(sizeof(double) * (size_t)p));
  for (const TiledMatrix* m : {&tp->x, &tp->xt}) {
    const size_t lb = lds_bytes(*m);
    if (lb > 160 * 1024)
      return fail(BBX_ERR_INVALID, "tile does not fit in LDS");
  }
  BBX_HIP(hipFuncSetAttribute(
      reinterpret_cast<const void*>(&tiled_spmv_kernel<false>),
      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  BBX_HIP(hipFuncSetAttribute(
      reinterpret_cast<const void*>(&tiled_spmv_kernel<true>),
      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
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
                        const double* rowscale, double* out, double* slab) {
  const unsigned grid = (unsigned)(m.n_panel * m.G);
  const size_t lb = lds_bytes(m);
  if (m.has_vals)
    hipLaunchKernelGGL(tiled_spmv_kernel<true>, dim3(grid), dim3(TILE_THREADS),
                       lb, h->stream, m.R, m.C, m.W, m.PR, m.G,
                       m.wg_tiles.as<int32_t>(), m.tiles.as<TileDesc>(),
                       m.slices.as<SliceMeta>(), m.rowids.as<uint16_t>(),
                       m.ids.as<uint2>(), m.vals.as<double>(), x, c_part, x0_ptr,
                       rowscale, out, slab);
  else
    hipLaunchKernelGGL(tiled_spmv_kernel<false>, dim3(grid),
                       dim3(TILE_THREADS), lb, h->stream, m.R, m.C, m.W, m.PR,
                       m.G, m.wg_tiles.as<int32_t>(), m.tiles.as<TileDesc>(),
                       m.slices.as<SliceMeta>(), m.rowids.as<uint16_t>(),
                       m.ids.as<uint2>(), nullptr, x, c_part, x0_ptr, rowscale,
                       out, slab);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

int launch_dot_tiled(bbx_design* h, const double* d_v,
                     const double* d_rowscale, double* d_t) {
  TiledPair* tp = static_cast<TiledPair*>(h->tiled);
  const TiledMatrix& m = tp->x;
  const double* x = d_v + h->intercept;
  const double* x0 = h->intercept ? d_v : nullptr;
  BBX_TRY(timer_begin(h, 0));
  if (m.G == 1) {
    BBX_TRY(launch_tiled(h, m, x, part_slot(h, PS_C), x0, d_rowscale, d_t,
                         nullptr));
  } else {
    BBX_TRY(launch_tiled(h, m, x, nullptr, nullptr, nullptr, nullptr,
                         m.slab.as<double>()));
    hipLaunchKernelGGL(tiled_dot_finalize_kernel, dim3(1024), dim3(256), 0,
                       h->stream, m.R, m.G, m.slab.as<double>(),
                       part_slot(h, PS_C), x0, d_rowscale, d_t);
    BBX_HIP(hipGetLastError());
  }
  BBX_TRY(timer_end(h, 0));
  return BBX_OK;
}

int launch_tdot_tiled(bbx_design* h, const double* d_w,
                      const double* d_sumw_part, const TdotEpilogue& ep,
                      double* d_out) {
  TiledPair* tp = static_cast<TiledPair*>(h->tiled);
  const TiledMatrix& m = tp->xt;
  BBX_TRY(timer_begin(h, 1));
  BBX_TRY(launch_tiled(h, m, d_w, nullptr, nullptr, nullptr, nullptr,
                       m.slab.as<double>()));
  const double* gfull = m.slab.as<double>();
  if (m.G > 1) {
    hipLaunchKernelGGL(tiled_slab_sum_kernel, dim3(NPART), dim3(256), 0,
                       h->stream, m.R, m.G, m.slab.as<double>(),
                       h->tiled_gfull.as<double>());
    BBX_HIP(hipGetLastError());
    gfull = h->tiled_gfull.as<double>();
  }
  BBX_TRY(timer_end(h, 1));
  return launch_tdot_finalize(h, gfull, d_sumw_part, ep, d_out);
}

int tiled_matvec_bytes(const bbx_design* h, int64_t* dot_bytes,
                       int64_t* tdot_bytes) {
  const TiledPair* tp = static_cast<const TiledPair*>(h->tiled);
  if (!tp) return fail(BBX_ERR_STATE, "tiled format not built");
  // bytes of the format actually read + vector in + vector out (+ slabs)
  *dot_bytes = tp->x.stream_bytes() + 8 * (h->P + h->n) +
               (tp->x.G > 1 ? 16 * tp->x.G * h->n : 0);
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
