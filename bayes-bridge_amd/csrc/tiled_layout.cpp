// Builder and CPU emulator of the LDS-tiled sparse layout.  Plain C++17 (no
// HIP): linked into libbbx.so by hipcc and built on its own with g++ for the
// CPU tests and the sanitizer runs (tests/test_tiled_layout_cpu.py).
#include "tiled_layout.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>

#include <sched.h>

namespace bbx {

// CPUs' worth of time the cgroup grants this process (cgroup v2 `cpu.max`,
// v1 `cpu.cfs_quota_us`), 0 when unlimited or unreadable.
static int cgroup_cpu_quota() {
  {
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string quota;
    long long period = 0;
    if (f && (f >> quota >> period) && quota != "max" && period > 0) {
      const long long q = atoll(quota.c_str());
      if (q > 0) return (int)std::max<long long>(1, (q + period - 1) / period);
    }
  }
  std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us");
  std::ifstream fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");
  long long q = 0, period = 0;
  if (fq && fp && (fq >> q) && (fp >> period) && q > 0 && period > 0)
    return (int)std::max<long long>(1, (q + period - 1) / period);
  return 0;
}

// Worker threads the builder may keep busy.  The GPU boxes this runs on show
// 256 cores and grant 16 (cgroup quota): threads beyond the quota are
// throttled, not scheduled (measured with the OpenMP baseline: 110 GB/s at 32
// threads, 0.6 GB/s at 256), and the ranks of a multi-GPU job build their
// layouts side by side on one host.  So: affinity mask, capped by the quota,
// shared evenly among the local ranks (LOCAL_WORLD_SIZE of
// torch.distributed.run), capped by `max_threads`.  BBX_BUILD_THREADS=N
// overrides all of it (N <= 0: the machine's core count, the old behaviour).
int cores_per_rank() {
  int n = 0;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (n < 1) n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  const int quota = cgroup_cpu_quota();
  if (quota > 0 && quota < n) n = quota;
  int ranks = 1;
  if (const char* e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
  return std::max(1, n / ranks);
}

int builder_threads(int max_threads) {
  if (const char* e = getenv("BBX_BUILD_THREADS")) {
    int n = atoi(e);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(n, std::max(1, max_threads)));
  }
  int n = 0;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (n < 1) n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  const int quota = cgroup_cpu_quota();
  if (quota > 0 && quota < n) n = quota;
  int ranks = 1;
  if (const char* e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
  n = std::max(1, n / ranks);
  return std::max(1, std::min(n, std::max(1, max_threads)));
}

TiledOptions TiledOptions::from_env(bool transpose) {
  TiledOptions o;
  auto geti = [&](const char* name, int* dst) {
    std::string key(name);
    const char* e = transpose ? getenv((key + "_T").c_str()) : nullptr;
    if (!e) e = getenv(name);
    if (e) *dst = atoi(e);
  };
  geti("BBX_TILED_PR", &o.force_PR);
  geti("BBX_TILED_G", &o.force_G);
  geti("BBX_TILED_BLOCKS", &o.force_blocks);
  geti("BBX_TILED_PACK", &o.packed);  // 0 / 1: plain ids / groups of five forced
  if (getenv("BBX_TILED_STATS")) o.stats = true;
  return o;
}

// ----------------------------------------------------------------- builder

struct PanelBuild {
  std::vector<Ids4> ids;
  std::vector<double> vals;
  std::vector<SliceMeta> slices;  // first_quad local to the panel
  std::vector<uint32_t> rowids;
  std::vector<TileDesc> tiles;    // slice ids local to the panel
  std::vector<int32_t> group_tile_count;
  std::vector<FoldDesc> folds;    // split rows of this panel
  int split_T = 0;
  std::vector<BatchDesc> descs;   // quad0/row_slot local to the panel
  std::vector<int32_t> wave_desc; // [G * TILE_WAVES] local start of each wave
  std::vector<uint32_t> wg_quad0; // [G] first step of each workgroup (panel-local)
  // schedule statistics (BBX_TILED_STATS): per workgroup, in batches
  std::vector<int64_t> wg_critical;  // sum over tiles of the busiest wave
  std::vector<int64_t> wg_total;     // all waves, all tiles
  int64_t dup_quads = 0;             // quads re-loaded to fill a batch
  // where the padding of the id stream comes from, in entry slots
  int64_t pad_round = 0;   // a row segment rounds up to whole steps (4 / 5 entries)
  int64_t pad_slice = 0;   // a slice is as long as its longest row, 128 rows wide
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
  std::vector<Ids4> new_ids;
  std::vector<double> new_vals;
  std::vector<uint32_t> new_rowids;
  std::vector<SliceMeta> new_slices;
  new_ids.reserve(pb.ids.size());
  new_rowids.reserve(pb.rowids.size());
  new_slices.reserve(pb.slices.size());
  if (has_vals) new_vals.reserve(pb.vals.size());
  pb.wg_quad0.assign((size_t)G, 0u);
  for (int g = 0; g < G; ++g) {
    const size_t t0 = tile_cursor, t1 = tile_cursor + pb.group_tile_count[g];
    tile_cursor = t1;
    // (a workgroup's steps are one contiguous stretch of the re-laid stream)
    const uint32_t wg_first = (uint32_t)(new_ids.size() / LANES);
    pb.wg_quad0[(size_t)g] = wg_first;
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
          sm.first_quad = (uint32_t)(new_ids.size() / LANES);
          sm.n_quad = old.n_quad;
          const uint32_t new_sl = (uint32_t)new_slices.size();
          new_slices.push_back(sm);
          const size_t src = (size_t)old.first_quad * LANES;
          const size_t cnt = (size_t)old.n_quad * LANES;
          new_ids.insert(new_ids.end(), pb.ids.begin() + src,
                         pb.ids.begin() + src + cnt);
          if (has_vals)
            new_vals.insert(new_vals.end(), pb.vals.begin() + src * 8,
                            pb.vals.begin() + (src + cnt) * 8);
          new_rowids.insert(new_rowids.end(),
                            pb.rowids.begin() + (size_t)sl * LANES,
                            pb.rowids.begin() + (size_t)(sl + 1) * LANES);
          for (uint32_t q0 = 0; q0 < sm.n_quad; q0 += (uint32_t)batch) {
            BatchDesc d;
            d.quad0 = sm.first_quad + q0 - wg_first;   // workgroup-relative
            d.row_slot = new_sl * LANES;
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
  int32_t steps = 0;    // steps this row needs (sort key)
  int32_t g_begin = 0;  // packed layout: first group in the tile's group list
};

// Groups of one (chunk of a) row in the packed layout (tiled_layout.hpp,
// packed_slot): appended to *out when given; returns their number.  `Wl` is
// the terminal zero slot of the slice.
static int pack_groups(const int32_t* colidx, int32_t begin, int32_t len,
                       int64_t col0, int Wl, std::vector<uint64_t>* out) {
  int n = 0;
  int32_t i = 0;
  while (i < len) {
    int32_t prev = packed_slot((int)(colidx[begin + i] - col0));
    uint64_t g = (uint64_t)prev;
    int k = 1;
    while (k < 5 && i + k < len) {
      const int32_t cur = packed_slot((int)(colidx[begin + i + k] - col0));
      const int32_t d = cur - prev;
      if (d < 0 || d > 4095) break;  // far apart (or unsorted): new group
      g |= (uint64_t)d << (14 + 12 * (k - 1));
      prev = cur;
      ++k;
    }
    if (k < 5) {  // step onto the next zero slot and stay there
      int32_t z = prev | 4095;
      if (z > Wl) z = Wl;
      g |= (uint64_t)(z - prev) << (14 + 12 * (k - 1));
    }
    if (out) out->push_back(g);
    i += k;
    ++n;
  }
  return n;
}

// Bank-aware entry order of one 32-lane half of a slice.
//
// A ds_read_b64 of the kernel gathers entry k of the rows of 32 lanes at once;
// lanes whose ids fall into the same LDS bank pair ((id mod 32), 8-byte slots)
// but differ serialise.  With ids in ascending order inside a row the banks
// are random: ~3.5 LDS cycles per gather instead of 1 (measured: 64 % of the
// LDS cycles of the kernel were bank conflicts).  The order of the entries
// INSIDE a row is free (it only fixes the order of the additions), so the
// builder deals them to the positions k = 0, 1, ... such that the 32 lanes hit
// distinct banks wherever it can: rows with the fewest banks left choose
// first, a row prefers (1) an id another lane already reads at this k
// (same address: broadcast), then (2) a free bank in which it holds the most
// entries.  Costs nothing at run time; the result stays deterministic.
//   rows[i]  = {pointer to the row's block-local ids, length}, i < n_rows <= 32
//   perm[i]  = for every position k the index of the entry placed there
struct HalfRow {
  const int32_t* col;  // colidx + begin
  int32_t len;
};
struct BankScratch {
  std::vector<int32_t> ent;     // entry indices of all rows, bucketed by bank
  std::vector<int32_t> top;     // [row * 32 + bank]: one past the bucket's last
  std::vector<int32_t> bot;     // [row * 32 + bank]: bucket's first
};
// `nb` = number of bank slots the lanes of one LDS group compete for: 32
// eight-byte slots for the ds_read_b64 of the single-chain kernel (groups of 32
// lanes), 16 sixteen-byte slots for the ds_read_b128 of the batched kernels
// (groups of 16 lanes, MI355X_MICROARCH.md "LDS").
static void bank_aware_order(const HalfRow* rows, int n_rows, int64_t col0,
                             int pad_bank, const uint8_t* hot, BankScratch& sc,
                             std::vector<int32_t>* const* perm, int nb) {
  const int bmask = nb - 1;
  int max_len = 0;
  size_t total = 0;
  for (int i = 0; i < n_rows; ++i) {
    max_len = std::max(max_len, rows[i].len);
    total += (size_t)rows[i].len;
  }
  sc.ent.resize(total);
  sc.top.assign((size_t)n_rows * 32, 0);  // (stride 32 for either nb)
  sc.bot.assign((size_t)n_rows * 32, 0);
  int left[32], n_banks[32];
  uint32_t avail[32];  // banks in which the row still holds entries
  {  // counting sort of every row's entries by bank; inside a bucket the
     // entries descend, so that taking from the top yields ascending ids
    size_t off = 0;
    for (int i = 0; i < n_rows; ++i) {
      int32_t cnt[33] = {0};
      const int32_t len = rows[i].len;
      for (int32_t k = 0; k < len; ++k) cnt[((rows[i].col[k] - col0) & bmask) + 1] += 1;
      int nbk = 0;
      for (int b = 0; b < 32; ++b) {
        nbk += cnt[b + 1] > 0;
        cnt[b + 1] += cnt[b];
        sc.bot[(size_t)i * 32 + (size_t)b] = (int32_t)off + cnt[b];
        sc.top[(size_t)i * 32 + (size_t)b] = (int32_t)off + cnt[b];
      }
      for (int32_t k = len - 1; k >= 0; --k) {
        const int b = (int)((rows[i].col[k] - col0) & bmask);
        sc.ent[(size_t)sc.top[(size_t)i * 32 + (size_t)b]++] = k;
      }
      left[i] = len;
      n_banks[i] = nbk;
      avail[i] = 0;
      for (int b = 0; b < 32; ++b)
        if (cnt[b + 1] > cnt[b]) avail[i] |= 1u << b;
      perm[i]->resize((size_t)len);
      off += (size_t)len;
    }
  }
  int order[32];
  for (int k = 0; k < max_len; ++k) {
    uint32_t used = 0, used_hot = 0;
    int64_t used_id[32];
    // most constrained first: fewest distinct banks left (counting sort)
    int head[34] = {0};
    int n_act = 0;
    for (int i = 0; i < n_rows; ++i) {
      if (left[i] > 0) {
        head[n_banks[i] + 1] += 1;
        ++n_act;
      } else if (rows[i].len <= k) {
        used |= 1u << pad_bank;  // this lane reads the 0.0 at xs[W]
      }
    }
    for (int b = 0; b < 33; ++b) head[b + 1] += head[b];
    for (int i = 0; i < n_rows; ++i)
      if (left[i] > 0) order[head[n_banks[i]]++] = i;
    for (int a = 0; a < n_act; ++a) {
      const int i = order[a];
      int32_t* top = &sc.top[(size_t)i * 32];
      const int32_t* bot = &sc.bot[(size_t)i * 32];
      int pick_bank = -1;
      int32_t pick_slot = -1;
      // (1) the same address as a lane already placed at this k (a broadcast);
      // only columns frequent enough in this tile to recur inside 32 rows
      for (uint32_t mset = used_hot & avail[i]; mset && pick_bank < 0;
           mset &= mset - 1) {
        const int b = __builtin_ctz(mset);
        for (int32_t t = bot[b]; t < top[b]; ++t)
          if (rows[i].col[sc.ent[(size_t)t]] - col0 == used_id[b]) {
            pick_bank = b;
            pick_slot = t;
            break;
          }
      }
      if (pick_bank >= 0) {
        // remove slot t from the bucket: move the top entry into it
        const int32_t e = sc.ent[(size_t)pick_slot];
        sc.ent[(size_t)pick_slot] = sc.ent[(size_t)(top[pick_bank] - 1)];
        top[pick_bank] -= 1;
        (*perm[i])[(size_t)k] = e;
      } else {
        // (2) a free bank, the one this row holds the most entries in;
        // (3) none free: its fullest bucket (a conflict either way)
        int32_t best = 0;
        uint32_t cand = avail[i] & ~used;
        if (!cand) cand = avail[i];
        for (uint32_t mset = cand; mset; mset &= mset - 1) {
          const int b = __builtin_ctz(mset);
          const int32_t c = top[b] - bot[b];
          if (c > best) {
            best = c;
            pick_bank = b;
          }
        }
        const int32_t e = sc.ent[(size_t)(top[pick_bank] - 1)];
        top[pick_bank] -= 1;
        (*perm[i])[(size_t)k] = e;
        if (!(used & (1u << pick_bank))) {
          used |= 1u << pick_bank;
          const int64_t id = rows[i].col[e] - col0;
          used_id[pick_bank] = id;
          if (hot && hot[id]) used_hot |= 1u << pick_bank;
        }
      }
      if (top[pick_bank] == bot[pick_bank]) {
        n_banks[i] -= 1;
        avail[i] &= ~(1u << pick_bank);
      }
      left[i] -= 1;
    }
  }
}

static void build_panel(int64_t R, int64_t C, const int64_t* rowptr,
                        const int32_t* colidx_all, const double* vals_all, int W,
                        int n_block, int PR, int G, int extra_budget,
                        int panel, bool packed, const TiledOptions& opt,
                        PanelBuild& pb) {
  const int Wl = packed_slots(W);  // packed: terminal zero slot of a slice
  std::vector<uint64_t> groups;    // packed: groups of the tile's rows
  const int64_t row0 = (int64_t)panel * PR;
  const int rows_here = (int)std::min<int64_t>(PR, R - row0);
  // Entry positions below are RELATIVE to the panel's first entry: a panel of
  // <= TILE_PR_MAX rows holds far fewer than 2^31 entries, the matrix may hold
  // more (64-bit row pointers, bbx_design_create_csr64).
  const int64_t base = rowptr[row0];
  const int32_t* colidx = colidx_all + base;
  const double* vals = vals_all ? vals_all + base : nullptr;
  // pass 1: segment of every row in every column block
  std::vector<int32_t> seg_begin((size_t)rows_here * n_block),
      seg_len((size_t)rows_here * n_block);
  std::vector<int32_t> max_seg(rows_here, 0);
  for (int r = 0; r < rows_here; ++r) {
    int32_t k = (int32_t)(rowptr[row0 + r] - base);
    const int32_t e = (int32_t)(rowptr[row0 + r + 1] - base);
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
      const bool t_env = opt.t_factor > 0.;
      if (t_env) t_factor = opt.t_factor;
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
      // Balance: a slice of T-entry chunks takes T / 4 steps on ONE wave while
      // the whole tile is densest_entries / 512 wave-steps for 16 waves; a
      // chunk longer than the tile's fair share per wave is its critical path.
      if (!t_env && opt.chains > 1) {
        int t_bal = (int)(4 * ((densest_entries + 512 * TILE_WAVES - 1) /
                               (512 * TILE_WAVES)));
        if (t_bal < 8) t_bal = 8;
        if (t_bal < t_min) t_min = t_bal;
      }
      const int64_t want_rows = 2 * TILE_WAVES * SLICE_ROWS;
      if (densest_rows < (3 * TILE_WAVES * SLICE_ROWS) / 2 && !t_env) {
        // (whole steps: 4 entries, or a packed group of 5)
        const int gran = packed ? 5 : 4;
        int t_par = (int)((densest_entries + want_rows - 1) / want_rows);
        t_par = (t_par + gran - 1) / gran * gran;
        if (t_par < 2 * gran) t_par = 2 * gran;
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
  std::vector<std::vector<int32_t>> slice_perm;  // bank-aware entry order
  BankScratch bank_scratch;
  std::vector<uint8_t> hot;        // per tile: column recurs within 32 rows
  std::vector<uint16_t> col_rows;  // per tile: rows holding the column
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
    if (opt.bank_aware && !packed) {
      // columns present in >= 1/16 of the tile's rows: likely to appear twice
      // among the 32 rows a gather instruction serves
      col_rows.assign((size_t)W + 1, 0);
      for (const VRow& v : vrows)
        for (int32_t k = 0; k < v.len; ++k) {
          uint16_t& c = col_rows[(size_t)(colidx[v.begin + k] - col0)];
          if (c < 0xFFFF) ++c;
        }
      hot.assign((size_t)W + 1, 0);
      const size_t thresh = std::max<size_t>(2, vrows.size() / 256);
      for (int j = 0; j <= W; ++j) hot[(size_t)j] = col_rows[(size_t)j] >= thresh;
    }
    // sort key: steps the row needs (4 entries, or one packed group, per step)
    int max_key = 0;
    groups.clear();
    for (VRow& v : vrows) {
      if (packed) {
        v.g_begin = (int32_t)groups.size();
        v.steps = pack_groups(colidx, v.begin, v.len, col0, Wl, &groups);
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
      {
        const int per_step = packed ? 5 : 4;
        int64_t steps_rows = 0;
        for (int i = 0; i < rows_in; ++i) {
          steps_rows += sorted[base + i].steps;
          pb.pad_round += (int64_t)sorted[base + i].steps * per_step -
                          sorted[base + i].len;
        }
        pb.pad_slice += ((int64_t)nq * SLICE_ROWS - steps_rows) * per_step;
      }
      SliceMeta sm;
      sm.first_quad = (uint32_t)(pb.ids.size() / LANES);
      sm.n_quad = nq;
      pb.slices.push_back(sm);
      const size_t id0 = pb.ids.size();
      pb.ids.resize(id0 + (size_t)nq * LANES);
      if (vals) pb.vals.resize((id0 + (size_t)nq * LANES) * 8, 0.);
      // entry order inside the rows (see bank_aware_order): one problem per
      // (row A | row B) x (lanes 0-31 | lanes 32-63)
      // (a packed group holds its entries in ascending order: no dealing)
      const bool reorder = opt.bank_aware && !packed;
      if (reorder) {
        slice_perm.resize(SLICE_ROWS);
        // lanes whose gathers are served in the same LDS cycle(s): the two
        // 32-lane halves for ds_read_b64, four 16-lane groups for ds_read_b128
        static const int kGroups128[4][16] = {
            {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
            {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
            {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
            {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
        const bool wide_reads = opt.chains > 1;
        const int n_groups = wide_reads ? 4 : 2, group_lanes = wide_reads ? 16 : 32;
        const int nb = wide_reads ? 16 : 32;
        for (int side = 0; side < 2; ++side)
          for (int grp = 0; grp < n_groups; ++grp) {
            HalfRow hr[32];
            std::vector<int32_t>* pp[32];
            int n_hr = 0;
            for (int i = 0; i < group_lanes; ++i) {
              const int lane = wide_reads ? kGroups128[grp][i] : grp * 32 + i;
              const int idx = base + side * LANES + lane;
              if (idx >= base + rows_in || idx >= n_rows) continue;
              hr[n_hr].col = colidx + sorted[idx].begin;
              hr[n_hr].len = sorted[idx].len;
              pp[n_hr] = &slice_perm[(size_t)(side * LANES + lane)];
              ++n_hr;
            }
            if (n_hr > 0)
              bank_aware_order(hr, n_hr, col0, W & (nb - 1), hot.data(),
                               bank_scratch, pp, nb);
          }
      }
      if (packed && opt.bank_aware) {
        // Groups ascend inside, but WHICH group a row presents at step q is
        // free (it only fixes the order of the additions).  Per 32-lane half
        // and row side, step by step: every lane takes, among the groups it has
        // not placed yet, the one whose five slots fall into the LDS banks the
        // lanes before it have loaded least at their positions.
        for (int side = 0; side < 2; ++side)
          for (int half32 = 0; half32 < 2; ++half32) {
            int occ[5][32];
            for (uint32_t q = 0; q < nq; ++q) {
              memset(occ, 0, sizeof(occ));
              for (int i = 0; i < 32; ++i) {
                const int idx = base + side * LANES + half32 * 32 + i;
                if (idx >= base + rows_in || idx >= n_rows) break;
                const VRow& v = sorted[idx];
                if ((int)q >= v.steps) continue;
                uint64_t* gr = &groups[(size_t)v.g_begin];
                int best = (int)q, best_cost = 1 << 30;
                for (int c = (int)q; c < v.steps; ++c) {
                  const uint64_t g = gr[c];
                  uint32_t sl = (uint32_t)(g & 0x3FFFu);
                  int cost = occ[0][sl & 31];
                  for (int u = 1; u < 5; ++u) {
                    sl += (uint32_t)((g >> (14 + 12 * (u - 1))) & 0xFFFu);
                    cost += occ[u][sl & 31];
                  }
                  if (cost < best_cost) {
                    best_cost = cost;
                    best = c;
                  }
                }
                std::swap(gr[q], gr[best]);
                const uint64_t g = gr[q];
                uint32_t sl = (uint32_t)(g & 0x3FFFu);
                occ[0][sl & 31] += 1;
                for (int u = 1; u < 5; ++u) {
                  sl += (uint32_t)((g >> (14 + 12 * (u - 1))) & 0xFFFu);
                  occ[u][sl & 31] += 1;
                }
              }
            }
          }
      }
      for (int l = 0; l < LANES; ++l) {
        // lane l owns sorted rows base + l (A) and base + 64 + l (B)
        const VRow* vr[2] = {nullptr, nullptr};
        if (l < rows_in) vr[0] = &sorted[base + l];
        if (LANES + l < rows_in) vr[1] = &sorted[base + LANES + l];
        pb.rowids.push_back((uint32_t)(vr[0] ? vr[0]->slot : NO_ROW) |
                            ((uint32_t)(vr[1] ? vr[1]->slot : NO_ROW) << 16));
        for (uint32_t q = 0; q < nq && packed; ++q) {
          uint64_t g[2];
          for (int half = 0; half < 2; ++half) {
            const VRow* v = vr[half];
            g[half] = (v && (int)q < v->steps)
                          ? groups[(size_t)v->g_begin + q]
                          : (uint64_t)Wl;  // empty: five times the zero slot
          }
          Ids4 step;
          step.x = (uint32_t)g[0];
          step.y = (uint32_t)(g[0] >> 32);
          step.z = (uint32_t)g[1];
          step.w = (uint32_t)(g[1] >> 32);
          pb.ids[id0 + (size_t)q * LANES + l] = step;
        }
        for (uint32_t q = 0; q < nq && !packed; ++q) {
          uint16_t e[8];
          for (int half = 0; half < 2; ++half) {
            const VRow* v = vr[half];
            for (int u = 0; u < 4; ++u) {
              const int k = (int)q * 4 + u;
              if (v && k < v->len) {
                const int32_t src =
                    v->begin +
                    (reorder ? slice_perm[(size_t)(half * LANES + l)][(size_t)k] : k);
                e[half * 4 + u] = (uint16_t)(colidx[src] - col0);
                if (vals)
                  pb.vals[(id0 + (size_t)q * LANES + l) * 8 + half * 4 + u] =
                      vals[src];
              } else {
                e[half * 4 + u] = (uint16_t)W;  // xs[W] == 0
              }
            }
          }
          Ids4 step;
          step.x = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
          step.y = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
          step.z = (uint32_t)e[4] | ((uint32_t)e[5] << 16);
          step.w = (uint32_t)e[6] | ((uint32_t)e[7] << 16);
          pb.ids[id0 + (size_t)q * LANES + l] = step;
        }
      }
    }
    td.slice_end = (int32_t)pb.slices.size();
    pb.tiles.push_back(td);
    pb.group_tile_count[cb / blocks_per_group] += 1;
  }
  build_schedules(pb, G, vals ? BATCH_VAL : BATCH_BIN);
}

// LDS doubles one right-hand side may use: slice + 8 + accumulators + extras
// (2 KB of the CU's LDS stay free for static use).
static int lds_budget_per_chain(int K) {
  return (int)((TILE_LDS_BYTES - 2048) / 8) / K;
}

// Picks (PR, G): row panels x groups of column blocks.  One workgroup runs per
// CU (it owns the CU's LDS), so the launch should be a single round of <= 256
// workgroups of equal work.  Cost model fitted on MI355X (profiles/,
// LABNOTES.md 2.1): a tile costs ~4.3 us of fixed time (slice refill from L2, two
// barriers, pipeline ramp) plus ~24 ps per stored entry streamed.
static double shape_cost(int64_t R, int64_t nnz, int n_block, int pr, int g,
                         int K) {
  const int64_t n_panel = (R + pr - 1) / pr;
  const int bpg = (n_block + g - 1) / g;
  const double n_wg = (double)n_panel * g;
  const double rounds = std::ceil(n_wg / (double)TILE_WG_PER_ROUND);
  const double rows = (double)std::min<int64_t>(pr, R);
  const double tile_nnz = (double)nnz * rows / (double)R / n_block;
  const double per_tile = 4.3 + tile_nnz * 24e-6;            // us
  double cost = rounds * bpg * per_tile + 6.;
  if (g > 1) cost += (double)R * g * K * 16. / 4e6;          // slab pass
  return cost;
}

static void choose_shape(int64_t R, int64_t C, int64_t nnz, int n_block, int W,
                         int K, bool heavy_rows, int* PR_out, int* G_out,
                         double* cost_out) {
  double best = 1e300;
  int best_pr = 128, best_g = 1;
  // (K = 1: four more slots, the zero slots of a packed slice)
  const int lds_rows = lds_budget_per_chain(K) - (W + 8) - (K == 1 ? 4 : 0);
  // (panels beyond 4096 rows only through BBX_TILED_PR[_T]: at 1M x 50k the
  // X^T geometries PR = 6272 x 96 blocks x G = 32 and PR = 5056 x 75 x 25 ran
  // the main kernel in 46.9 / 49.7 us against 48.5 us, with twice the slab
  // traffic for the epilogue kernel -- profiles/r02_ab_geometry.txt)
  int pr_cap = TILE_PR_MAX / K < 4096 ? TILE_PR_MAX / K : 4096;
  // Room for the extra accumulators of split rows.  Batched layouts (K > 1)
  // have small tiles -- a handful of steps per wave -- where ONE long slice is
  // the tile's critical path: their rows must be cut short, which takes
  // accumulators.  Measured on the 1M x 50k design, X^T, K = 2: 312 extras ->
  // split threshold 53, busiest wave 128 batches against 57.6 ideal (X^T W
  // 76 us); ~1400 extras -> threshold 9-21, busiest wave 70.
  // (only where the row lengths are heavy-tailed: balanced rows -- X itself
  // -- are not split and would only lose slice width)
  int reserve = (K > 1 && heavy_rows) ? std::max(256, lds_rows / 3) : 256;
  if (lds_rows - reserve < pr_cap) pr_cap = lds_rows - reserve;
  if (pr_cap < 128) pr_cap = 128;
  for (int pr = 128; pr <= pr_cap; pr += 128) {
    for (int g = 1; g <= n_block; ++g) {
      const int bpg = (n_block + g - 1) / g;
      if ((n_block + bpg - 1) / bpg != g) continue;  // not a distinct split
      const double cost = shape_cost(R, nnz, n_block, pr, g, K);
      if (cost < best) {
        best = cost;
        best_pr = pr;
        best_g = g;
      }
    }
  }
  *PR_out = best_pr;
  *G_out = best_g;
  if (cost_out) *cost_out = best;
}

// Geometry of one orientation for K right-hand sides: W, n_block, PR, G, n_panel
// and the cost model's estimate of one product, in microseconds.  Everything
// build_tiled_host decides before it touches the entries; also what
// tiled_model_cost answers from (batch-width decisions need no built layout).
static int choose_geometry(int64_t R, int64_t C, int64_t nnz,
                           const int64_t* rowptr, const TiledOptions& opt,
                           TiledHost& m, const char** why) {
  const int K = opt.chains;
  if (K != 1 && K != 2 && K != 4) {
    *why = "chains must be 1, 2 or 4";
    return -1;
  }
  m.K = K;
  auto width_for = [&](int n_block) {
    int64_t w = (C + n_block - 1) / n_block;
    return (int)((w + 63) / 64 * 64);
  };
  if (K == 1) {
    m.n_block = (int)((C + TILE_W_MAX - 1) / TILE_W_MAX);
    if (opt.force_blocks > m.n_block) m.n_block = opt.force_blocks;
    if (m.n_block < 1) m.n_block = 1;
    m.W = width_for(m.n_block);
    choose_shape(R, C, nnz, m.n_block, m.W, K, false, &m.PR, &m.G, nullptr);
  } else {
    // K slices and K accumulator sets share the LDS: the slice width is part
    // of the search (a wide slice means few tile switches but short panels,
    // i.e. more workgroups than CUs).  Candidates: every block count from the
    // widest slice that leaves room for 128 rows + 256 extras down to slices
    // a quarter as wide.
    // heavy-tailed row lengths (the columns of simulate_data.py designs as
    // rows of X^T): longest row > 6x the mean, the test build_panel applies
    int64_t longest = 0;
    for (int64_t r = 0; r < R; ++r)
      longest = std::max<int64_t>(longest, rowptr[r + 1] - rowptr[r]);
    const bool heavy_rows = R > 0 && (double)longest > 6. * (double)nnz / (double)R;
    const int w_cap = (lds_budget_per_chain(K) - 8 - 128 - 256) / 64 * 64;
    int nb_min = (int)((C + w_cap - 1) / w_cap);
    if (nb_min < 1) nb_min = 1;
    if (opt.force_blocks > nb_min) nb_min = opt.force_blocks;
    const int nb_max = opt.force_blocks > 0 ? nb_min : 4 * nb_min + 4;
    double best = 1e300;
    for (int nb = nb_min; nb <= nb_max; ++nb) {
      const int w = width_for(nb);
      if (nb > nb_min && w == width_for(nb - 1)) continue;
      int pr, g;
      double cost;
      choose_shape(R, C, nnz, nb, w, K, heavy_rows, &pr, &g, &cost);
      if (cost < best) {
        best = cost;
        m.n_block = nb;
        m.W = w;
        m.PR = pr;
        m.G = g;
      }
      if (w <= 64) break;
    }
  }
  if (opt.force_PR > 0) m.PR = opt.force_PR;
  if (opt.force_G > 0) m.G = opt.force_G;
  if (m.PR < 64) m.PR = 64;
  if (m.PR > TILE_PR_MAX / K) m.PR = TILE_PR_MAX / K;
  if (m.G < 1) m.G = 1;
  if (m.G > m.n_block) m.G = m.n_block;
  {  // normalise G so that every group is non-empty
    const int bpg = (m.n_block + m.G - 1) / m.G;
    m.G = (m.n_block + bpg - 1) / bpg;
  }
  m.n_panel = (int)((R + m.PR - 1) / m.PR);
  m.model_cost_us = shape_cost(R, nnz, m.n_block, m.PR, m.G, K);
  return 0;
}

double tiled_model_cost(int64_t R, int64_t C, int64_t nnz,
                        const int64_t* rowptr, int chains) {
  TiledOptions opt;
  opt.chains = chains;
  TiledHost m;
  const char* why = nullptr;
  if (choose_geometry(R, C, nnz, rowptr, opt, m, &why) != 0) return -1.;
  return m.model_cost_us;
}

int build_tiled_host(int64_t R, int64_t C, int64_t nnz, const int64_t* rowptr,
                     const int32_t* colidx, const double* vals,
                     const TiledOptions& opt, TiledHost* out, std::string* err) {
  TiledHost& m = *out;
  m = TiledHost();
  auto fail = [&](const char* msg) {
    if (err) *err = msg;
    return -1;
  };
  m.R = R;
  m.C = C;
  m.nnz = nnz;
  m.has_vals = vals != nullptr;
  const int K = opt.chains;
  {
    const char* why = nullptr;
    if (choose_geometry(R, C, nnz, rowptr, opt, m, &why) != 0) return fail(why);
  }
  // LDS left after the vector slice and the row accumulators pays for the
  // extra accumulators of split rows (2 KB stay free for static LDS).
  int extra_budget = lds_budget_per_chain(K) - (m.W + 8) - m.PR - (K == 1 ? 4 : 0);
  if (extra_budget > 8192) extra_budget = 8192;
  if (extra_budget < 0) extra_budget = 0;
  if (opt.extra_budget >= 0) extra_budget = opt.extra_budget;

  std::vector<PanelBuild> pbs((size_t)m.n_panel);
  unsigned n_thr = (unsigned)builder_threads(opt.max_threads);
  if ((unsigned)m.n_panel < n_thr) n_thr = (unsigned)m.n_panel;
  // Packed groups or plain ids (value-free, one right-hand side)?  A row
  // segment of `len` entries takes ceil(len / 4) plain steps, or as many groups
  // as its gaps allow (five entries at best, fewer where columns lie more than
  // 4095 slots apart).  Groups cost ~40 % more index arithmetic per entry and
  // give up the bank-aware entry order, so they pay where the id stream comes
  // from HBM and the slices are long enough to hide the decode behind it.
  // Measured on MI355X (profiles/r04_packed_ids.txt; X~ v / X~^T w, us):
  //   1M x 50k, 100 / row   235 MB of plain ids  46.4 -> 43.8 / 49.7 -> 45.7
  //   400k x 20k, 100 / row 104 MB               32.1 -> 30.0 / 23.2 -> 21.6
  //   1M x 20k, 40 / row    103 MB               24.4 -> 22.7 / 25.4 -> 24.6
  //   1M x 50k, 50 / row    133 MB (12 / segment) 32.5 -> 32.1 / 32.5 -> 33.8
  //   200k x 20k             56 MB               25.8 -> 25.1 / 15.8 -> 16.6
  //   100k x 10k             26 MB               13.6 -> 14.8 / 12.5 -> 13.7
  // Hence: groups if they save >= 5 % of the steps, the plain ids would exceed
  // 80 MB and a row segment holds >= 20 entries on average.  One pass over the
  // entries; TiledOptions::packed = 0 / 1 overrides (BBX_TILED_PACK).
  m.packed = false;
  if (!vals && K == 1 && opt.packed != 0 && m.W + 4 <= (1 << 14)) {
    m.packed = opt.packed > 0;
    if (opt.packed < 0 && nnz > 0) {
      std::vector<int64_t> plain(n_thr, 0), grouped(n_thr, 0), segs(n_thr, 0);
      std::vector<std::thread> counters;
      const int Wl = packed_slots(m.W);
      for (unsigned t = 0; t < n_thr; ++t)
        counters.emplace_back([&, t]() {
          const int64_t r0 = R * (int64_t)t / n_thr, r1 = R * (int64_t)(t + 1) / n_thr;
          int64_t a = 0, b = 0, c = 0;
          for (int64_t r = r0; r < r1; ++r) {
            int64_t k = rowptr[r];
            const int64_t e = rowptr[r + 1];
            while (k < e) {
              const int64_t cb = colidx[k] / m.W;
              const int64_t col_end = (cb + 1) * m.W;
              const int64_t b0 = k;
              while (k < e && colidx[k] < col_end) ++k;
              a += (k - b0 + 3) / 4;
              b += pack_groups(colidx + b0, 0, (int32_t)(k - b0), cb * m.W, Wl,
                               nullptr);
              ++c;
            }
          }
          plain[t] = a;
          grouped[t] = b;
          segs[t] = c;
        });
      for (auto& th : counters) th.join();
      int64_t a = 0, b = 0, c = 0;
      for (unsigned t = 0; t < n_thr; ++t) {
        a += plain[t];
        b += grouped[t];
        c += segs[t];
      }
      // (a lane's step is 8 bytes per row: plain id bytes ~ 8 a)
      m.packed = (double)b < 0.95 * (double)a && 8 * a >= (int64_t)80e6 &&
                 nnz >= 20 * c;
    }
  }
  m.Wl = m.packed ? packed_slots(m.W) : m.W;
  const bool packed = m.packed;
  std::vector<std::thread> pool;
  std::vector<int> thread_status(n_thr, 0);
  for (unsigned t = 0; t < n_thr; ++t)
    pool.emplace_back([&, t]() {
      // an exception must not leave a worker thread (std::terminate)
      try {
        for (int p = (int)t; p < m.n_panel; p += (int)n_thr)
          build_panel(R, C, rowptr, colidx, vals, m.W, m.n_block, m.PR, m.G,
                      extra_budget, p, packed, opt, pbs[(size_t)p]);
      } catch (...) {
        thread_status[t] = -1;
      }
    });
  for (auto& th : pool) th.join();
  for (int st_t : thread_status)
    if (st_t < 0) return fail("out of host memory while tiling");

  if (opt.stats) {
    int64_t crit_max = 0, total = 0, dup = 0, quads = 0, n_wg = 0, crit_sum = 0;
    int stat_extra = 0, stat_T = 0;
    int64_t pad_round = 0, pad_slice = 0;
    for (auto& pb : pbs) {
      pad_round += pb.pad_round;
      pad_slice += pb.pad_slice;
      for (size_t g = 0; g < pb.wg_critical.size(); ++g) {
        crit_max = std::max(crit_max, pb.wg_critical[g]);
        crit_sum += pb.wg_critical[g];
        total += pb.wg_total[g];
        ++n_wg;
      }
      dup += pb.dup_quads;
      quads += (int64_t)(pb.ids.size() / LANES);
      int ex = 0;
      for (const FoldDesc& fd : pb.folds) ex += fd.count;
      stat_extra = std::max(stat_extra, ex);
      if (pb.split_T > 0 && (stat_T == 0 || pb.split_T < stat_T))
        stat_T = pb.split_T;
    }
    char line[1536];
    snprintf(line, sizeof(line),
            "[bbx tiled %lldx%lld K=%d] W=%d blocks=%d PR=%d G=%d split T=%d "
            "extras=%d workgroups=%lld: "
            "quads=%lld (+%lld re-loaded to fill batches, %.1f%%); batches per "
            "wave: ideal %.1f, mean critical path %.1f, worst workgroup %lld "
            "(%.1f%% over ideal); padding of the id stream: %.1f%% = %.1f%% rows "
            "rounded up to whole steps + %.1f%% slices as long as their longest "
            "row\n",
            (long long)R, (long long)C, K, m.W, m.n_block, m.PR, m.G, stat_T,
            stat_extra, (long long)n_wg, (long long)quads, (long long)dup,
            100. * (double)dup / (double)std::max<int64_t>(quads, 1),
            (double)total / (double)(n_wg * TILE_WAVES),
            (double)crit_sum / (double)n_wg, (long long)crit_max,
            100. * ((double)crit_max * n_wg * TILE_WAVES / (double)total - 1.),
            100. * (double)(pad_round + pad_slice) /
                (double)std::max<int64_t>(quads * SLICE_ROWS * (packed ? 5 : 4), 1),
            100. * (double)pad_round /
                (double)std::max<int64_t>(quads * SLICE_ROWS * (packed ? 5 : 4), 1),
            100. * (double)pad_slice /
                (double)std::max<int64_t>(quads * SLICE_ROWS * (packed ? 5 : 4), 1));
    m.stats = line;
    fputs(line, stderr);
  }
  // concatenate with offset fix-ups
  size_t tot_ids = 0, tot_slices = 0, tot_tiles = 0, tot_descs = 0;
  for (auto& pb : pbs) {
    tot_ids += pb.ids.size();
    tot_slices += pb.slices.size();
    tot_tiles += pb.tiles.size();
    tot_descs += pb.descs.size();
  }
  if (tot_ids / LANES >= ((size_t)1 << 32))
    return fail("matrix too large for the tiled format");
  std::vector<Ids4> ids(tot_ids);
  std::vector<double> vv(m.has_vals ? tot_ids * 8 : 0);
  std::vector<BatchDesc> descs(tot_descs);
  std::vector<int32_t> wave_desc((size_t)m.n_panel * m.G * TILE_WAVES, 0);
  std::vector<uint32_t> wg_quad0((size_t)m.n_panel * m.G, 0u);
  std::vector<uint32_t> rowids(tot_slices * LANES);
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
      memcpy(&ids[id_off], pb.ids.data(), pb.ids.size() * sizeof(Ids4));
    if (m.has_vals && !pb.vals.empty())
      memcpy(&vv[id_off * 8], pb.vals.data(), pb.vals.size() * sizeof(double));
    for (size_t k = 0; k < pb.descs.size(); ++k) {
      BatchDesc d = pb.descs[k];
      // (quad0 stays relative to the workgroup's first step)
      if (d.info & 15u) d.row_slot += (uint32_t)(sl_off * LANES);
      descs[de_off + k] = d;
    }
    for (size_t g = 0; g < pb.wg_quad0.size(); ++g)
      wg_quad0[(size_t)p * m.G + g] = pb.wg_quad0[g] + (uint32_t)(id_off / LANES);
    for (size_t k = 0; k < pb.wave_desc.size(); ++k)
      wave_desc[(size_t)p * m.G * TILE_WAVES + k] =
          pb.wave_desc[k] + (int32_t)de_off;
    if (!pb.rowids.empty())
      memcpy(&rowids[sl_off * LANES], pb.rowids.data(),
             pb.rowids.size() * sizeof(uint32_t));
    id_off += pb.ids.size();
    sl_off += pb.slices.size();
    de_off += pb.descs.size();
    std::vector<Ids4>().swap(pb.ids);
    std::vector<double>().swap(pb.vals);
  }
  panel_fold[(size_t)m.n_panel] = (int32_t)folds.size();
  m.n_quad = (int64_t)(tot_ids / LANES);
  m.n_slice = (int64_t)tot_slices;
  m.n_tile = (int64_t)tot_tiles;
  m.n_desc = (int64_t)tot_descs;
  // The kernel addresses the id and value streams with 32-bit byte offsets
  // INSIDE a workgroup's stretch (64-bit base per workgroup, wg_quad0): what
  // must stay below 4 GiB is one workgroup's share, not the stream.
  {
    size_t longest = 0;
    for (size_t k = 0; k < wg_quad0.size(); ++k) {
      const size_t end = k + 1 < wg_quad0.size() ? (size_t)wg_quad0[k + 1]
                                                 : tot_ids / LANES;
      longest = std::max(longest, end - (size_t)wg_quad0[k]);
    }
    if ((uint64_t)longest * LANES * (m.has_vals ? 64u : 16u) >= ((uint64_t)1 << 32))
      return fail("matrix too large for the tiled format");
  }
  // (row ids: 32-bit byte offsets from the start of the array)
  if (tot_slices * LANES >= ((size_t)1 << 30) || tot_descs >= ((size_t)1 << 31))
    return fail("matrix too large for the tiled format");
  {  // equal-stride schedules when the padding stays small
    const size_t n_wave = wave_desc.size();
    size_t max_len = 0;
    for (size_t k = 0; k < n_wave; ++k) {
      const size_t end = (k + 1 < n_wave) ? (size_t)wave_desc[k + 1] : tot_descs;
      max_len = std::max(max_len, end - (size_t)wave_desc[k]);
    }
    const size_t stride = (max_len + LANES - 1) / LANES * LANES;
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
    descs.resize(descs.size() + 2 * LANES, endd);
  }
  if ((size_t)m.lds_doubles() * sizeof(double) > (size_t)TILE_LDS_BYTES)
    return fail("tile does not fit in LDS");
  m.ids.swap(ids);
  m.vals.swap(vv);
  m.descs.swap(descs);
  m.wave_desc.swap(wave_desc);
  m.wg_quad0.swap(wg_quad0);
  m.rowids.swap(rowids);
  m.folds.swap(folds);
  m.panel_fold.swap(panel_fold);
  return 0;
}


// ---------------------------------------------------------------- emulator

namespace {

// The two accumulations of one lane and step, exactly as the kernel writes
// them (step_accumulate in spmv_tiled.hip).
inline void emu_step(const TiledHost& m, const double* xs, const Ids4& e,
                     const double* v, double& a0, double& a1, double& b0,
                     double& b1) {
  if (m.has_vals) {
    a0 += v[0] * xs[e.x & 0xFFFFu] + v[2] * xs[e.y & 0xFFFFu];
    a1 += v[1] * xs[e.x >> 16] + v[3] * xs[e.y >> 16];
    b0 += v[4] * xs[e.z & 0xFFFFu] + v[6] * xs[e.w & 0xFFFFu];
    b1 += v[5] * xs[e.z >> 16] + v[7] * xs[e.w >> 16];
  } else {
    a0 += xs[e.x & 0xFFFFu] + xs[e.y & 0xFFFFu];
    a1 += xs[e.x >> 16] + xs[e.y >> 16];
    b0 += xs[e.z & 0xFFFFu] + xs[e.w & 0xFFFFu];
    b1 += xs[e.z >> 16] + xs[e.w >> 16];
  }
}

// The packed step (step_accumulate_packed in spmv_tiled.hip): the five slots of
// row A's group and of row B's, gathered unconditionally.
inline void emu_step_packed(const double* xs, const Ids4& e, double& a0,
                            double& a1, double& b0, double& b1) {
  const uint32_t w[2][2] = {{e.x, e.y}, {e.z, e.w}};
  double* s0[2] = {&a0, &b0};
  double* s1[2] = {&a1, &b1};
  for (int h = 0; h < 2; ++h) {
    const uint64_t g = (uint64_t)w[h][0] | ((uint64_t)w[h][1] << 32);
    const uint32_t i0 = (uint32_t)(g & 0x3FFFu);
    const uint32_t i1 = i0 + (uint32_t)((g >> 14) & 0xFFFu);
    const uint32_t i2 = i1 + (uint32_t)((g >> 26) & 0xFFFu);
    const uint32_t i3 = i2 + (uint32_t)((g >> 38) & 0xFFFu);
    const uint32_t i4 = i3 + (uint32_t)((g >> 50) & 0xFFFu);
    *s0[h] += (xs[i0] + xs[i2]) + xs[i4];
    *s1[h] += xs[i1] + xs[i3];
  }
}

}  // namespace

void emulate_tiled_spmv(const TiledHost& m, const double* x,
                        std::vector<double>* slab) {
  slab->assign((size_t)m.G * (size_t)m.R, 0.);
  const int bpg = (m.n_block + m.G - 1) / m.G;
  const int n_acc = m.PR + m.n_extra;
  std::vector<double> xs((size_t)m.Wl + 8), acc((size_t)n_acc);
  struct WaveState {
    int64_t cursor;
    bool ended;
    double a0[LANES], a1[LANES], b0[LANES], b1[LANES];
  };
  std::vector<WaveState> ws(TILE_WAVES);
  for (int wg = 0; wg < m.n_panel * m.G; ++wg) {
    const int panel = wg / m.G, group = wg - panel * m.G;
    const int64_t row0 = (int64_t)panel * m.PR;
    const int rows_here = (int)std::min<int64_t>(m.PR, m.R - row0);
    std::fill(acc.begin(), acc.end(), 0.);
    for (int w = 0; w < TILE_WAVES; ++w) {
      WaveState& s = ws[w];
      s.cursor = m.desc_stride > 0
                     ? (int64_t)(wg * TILE_WAVES + w) * m.desc_stride
                     : m.wave_desc[(size_t)wg * TILE_WAVES + w];
      s.ended = false;
      for (int l = 0; l < LANES; ++l) s.a0[l] = s.a1[l] = s.b0[l] = s.b1[l] = 0.;
    }
    // tile by tile (the kernel's barriers): every wave processes its
    // descriptors of the tile, then all move on
    for (int cb = group * bpg;; ++cb) {
      bool any = false;
      for (int w = 0; w < TILE_WAVES; ++w)
        any = any || (!ws[w].ended &&
                      !(m.descs[(size_t)ws[w].cursor].info & BD_END));
      if (!any) break;
      const int64_t col0 = (int64_t)cb * m.W;
      const int cols_here =
          (int)std::max<int64_t>(0, std::min<int64_t>(m.W, m.C - col0));
      if (m.packed) {
        std::fill(xs.begin(), xs.end(), 0.);  // zero slots included
        for (int j = 0; j < cols_here; ++j)
          xs[(size_t)packed_slot(j)] = x[col0 + j];
      } else {
        for (int j = 0; j < m.W + 8; ++j)
          xs[(size_t)j] = j < cols_here ? x[col0 + j] : 0.;
      }
      for (int w = 0; w < TILE_WAVES; ++w) {
        WaveState& s = ws[w];
        if (s.ended) continue;
        bool first = true;
        for (;;) {
          const BatchDesc& d = m.descs[(size_t)s.cursor];
          if (d.info & BD_END) {
            s.ended = true;
            break;
          }
          if ((d.info & BD_TILE_FIRST) && !first) break;  // next tile
          first = false;
          const int cnt = (int)(d.info & 15u);
          for (int u = 0; u < cnt; ++u) {
            const size_t q = (size_t)m.wg_quad0[(size_t)wg] + (size_t)d.quad0 +
                             (size_t)u;
            for (int l = 0; l < LANES; ++l) {
              const size_t at = q * LANES + (size_t)l;
              if (m.packed)
                emu_step_packed(xs.data(), m.ids[at], s.a0[l], s.a1[l], s.b0[l],
                                s.b1[l]);
              else
                emu_step(m, xs.data(), m.ids[at],
                         m.has_vals ? &m.vals[at * 8] : nullptr, s.a0[l],
                         s.a1[l], s.b0[l], s.b1[l]);
            }
          }
          if (cnt > 0 && (d.info & BD_LAST)) {
            for (int l = 0; l < LANES; ++l) {
              const uint32_t rr = m.rowids[(size_t)d.row_slot + (size_t)l];
              const uint32_t ra = rr & 0xFFFFu, rb = rr >> 16;
              if (ra != NO_ROW) acc[ra] += s.a0[l] + s.a1[l];
              if (rb != NO_ROW) acc[rb] += s.b0[l] + s.b1[l];
              s.a0[l] = s.a1[l] = s.b0[l] = s.b1[l] = 0.;
            }
          }
          ++s.cursor;
        }
      }
    }
    for (int f = m.panel_fold[(size_t)panel]; f < m.panel_fold[(size_t)panel + 1];
         ++f) {
      const FoldDesc& fd = m.folds[(size_t)f];
      double v = acc[fd.row];
      for (int c = 0; c < fd.count; ++c) v += acc[(size_t)fd.first + c];
      acc[fd.row] = v;
    }
    double* dst = slab->data() + (size_t)group * (size_t)m.R + (size_t)row0;
    for (int r = 0; r < rows_here; ++r) dst[r] = acc[(size_t)r];
  }
}

double tiled_mean_gather_cycles(const TiledHost& m) {
  if (m.n_quad == 0) return 0.;
  // K == 1: ds_read_b64, two groups of 32 lanes, 32 eight-byte slots;
  // K > 1 : ds_read_b128 (one per plane of pairs), four groups of 16 lanes,
  //         16 sixteen-byte slots.  Equal addresses broadcast.
  static const int kGroups128[4][16] = {
      {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
      {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  const bool wide = m.K > 1;
  const int n_groups = wide ? 4 : 2, group_lanes = wide ? 16 : 32;
  const int bmask = wide ? 15 : 31;
  double cycles = 0., groups = 0.;
  if (m.packed) {
    // ten gathers per step (five slots of row A's group, five of row B's)
    for (int64_t q = 0; q < m.n_quad; ++q)
      for (int pos = 0; pos < 10; ++pos)
        for (int grp = 0; grp < 2; ++grp) {
          uint16_t seen[32][32];
          int n_seen[32] = {0};
          int worst = 1;
          for (int i = 0; i < 32; ++i) {
            const Ids4& e = m.ids[(size_t)q * LANES + (size_t)(grp * 32 + i)];
            const uint64_t g = pos < 5 ? ((uint64_t)e.x | ((uint64_t)e.y << 32))
                                       : ((uint64_t)e.z | ((uint64_t)e.w << 32));
            uint32_t id = (uint32_t)(g & 0x3FFFu);
            for (int k = 0; k < pos % 5; ++k)
              id += (uint32_t)((g >> (14 + 12 * k)) & 0xFFFu);
            const int bank = (int)(id & 31u);
            bool dup = false;
            for (int k = 0; k < n_seen[bank]; ++k) dup = dup || seen[bank][k] == id;
            if (!dup) {
              seen[bank][n_seen[bank]++] = (uint16_t)id;
              worst = std::max(worst, n_seen[bank]);
            }
          }
          cycles += worst;
          groups += 1.;
        }
    return cycles / groups;
  }
  for (int64_t q = 0; q < m.n_quad; ++q) {
    for (int pos = 0; pos < 8; ++pos)
      for (int grp = 0; grp < n_groups; ++grp) {
        uint16_t seen[32][32];
        int n_seen[32] = {0};
        int worst = 1;
        for (int i = 0; i < group_lanes; ++i) {
          const int l = wide ? kGroups128[grp][i] : grp * 32 + i;
          const Ids4& e = m.ids[(size_t)q * LANES + (size_t)l];
          const uint32_t word = pos < 2 ? e.x : pos < 4 ? e.y : pos < 6 ? e.z : e.w;
          // positions: x.lo x.hi y.lo y.hi z.lo z.hi w.lo w.hi
          const uint16_t id = (pos & 1) ? (uint16_t)(word >> 16) : (uint16_t)word;
          const int bank = id & bmask;
          bool dup = false;
          for (int k = 0; k < n_seen[bank]; ++k) dup = dup || seen[bank][k] == id;
          if (!dup) {
            seen[bank][n_seen[bank]++] = id;
            worst = std::max(worst, n_seen[bank]);
          }
        }
        cycles += worst;
        groups += 1.;
      }
  }
  return cycles / groups;
}

// ---- host-side CSR utilities of the 64-bit constructor ----------------------

int check_csr64_host(int64_t R, int64_t C, int64_t nnz, const int64_t* rowptr,
                     const int64_t* colidx, int max_threads) {
  if (rowptr[0] != 0 || rowptr[R] != nnz) return 1;
  unsigned n_thr = (unsigned)builder_threads(max_threads);
  if ((int64_t)n_thr > R) n_thr = (unsigned)std::max<int64_t>(R, 1);
  std::vector<int> bad(n_thr, 0);
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < n_thr; ++t)
    pool.emplace_back([&, t]() {
      const int64_t r0 = R * (int64_t)t / n_thr, r1 = R * (int64_t)(t + 1) / n_thr;
      int b = 0;
      for (int64_t r = r0; r < r1; ++r) {
        const int64_t k0 = rowptr[r], k1 = rowptr[r + 1];
        if (k0 < 0 || k1 < k0 || k1 > nnz) {
          b |= 1;
          continue;
        }
        int64_t prev = -1;
        for (int64_t k = k0; k < k1; ++k) {
          const int64_t c = colidx[k];
          if (c < 0 || c >= C) b |= 2;
          if (c < prev) b |= 4;
          prev = c;
        }
      }
      bad[t] = b;
    });
  for (auto& th : pool) th.join();
  int b = 0;
  for (int v : bad) b |= v;
  return b;
}

void transpose_csr_host(int64_t R, int64_t C, const int64_t* rowptr,
                        const int32_t* colidx, const double* vals,
                        int max_threads, HostCsr* out) {
  const int64_t nnz = rowptr[R];
  // per-thread column counts: n_thr * C counters, kept under ~2 GB
  unsigned n_thr = (unsigned)builder_threads(max_threads);
  while (n_thr > 1 && (double)n_thr * (double)C * 8. > 2e9) n_thr /= 2;
  if ((int64_t)n_thr > R) n_thr = (unsigned)std::max<int64_t>(R, 1);
  // thread t owns a contiguous range of rows with ~nnz / n_thr entries
  std::vector<int64_t> row_cut(n_thr + 1, R);
  row_cut[0] = 0;
  for (unsigned t = 1; t < n_thr; ++t) {
    const int64_t target = nnz / n_thr * t;
    row_cut[t] = std::lower_bound(rowptr, rowptr + R + 1, target) - rowptr;
    if (row_cut[t] > R) row_cut[t] = R;
    if (row_cut[t] < row_cut[t - 1]) row_cut[t] = row_cut[t - 1];
  }
  // (allocated here, not in the workers: a bad_alloc must reach the caller)
  std::vector<std::vector<int64_t>> cnt(n_thr);
  for (auto& c : cnt) c.assign((size_t)C, 0);
  {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_thr; ++t)
      pool.emplace_back([&, t]() {
        for (int64_t k = rowptr[row_cut[t]]; k < rowptr[row_cut[t + 1]]; ++k)
          cnt[t][(size_t)colidx[k]] += 1;
      });
    for (auto& th : pool) th.join();
  }
  // column j of X = row j of X^T: its entries in thread order are in row order
  out->rowptr.assign((size_t)C + 1, 0);
  int64_t at = 0;
  for (int64_t j = 0; j < C; ++j) {
    out->rowptr[(size_t)j] = at;
    for (unsigned t = 0; t < n_thr; ++t) {
      const int64_t c = cnt[t][(size_t)j];
      cnt[t][(size_t)j] = at;  // becomes thread t's write position in row j
      at += c;
    }
  }
  out->rowptr[(size_t)C] = at;
  out->colidx.resize((size_t)std::max<int64_t>(nnz, 1));
  out->vals.clear();
  if (vals) out->vals.resize((size_t)std::max<int64_t>(nnz, 1));
  {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_thr; ++t)
      pool.emplace_back([&, t]() {
        std::vector<int64_t>& pos = cnt[t];
        for (int64_t r = row_cut[t]; r < row_cut[t + 1]; ++r)
          for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
            const int64_t w = pos[(size_t)colidx[k]]++;
            out->colidx[(size_t)w] = (int32_t)r;
            if (vals) out->vals[(size_t)w] = vals[k];
          }
      });
    for (auto& th : pool) th.join();
  }
}

}  // namespace bbx

#ifdef BBX_LAYOUT_CAPI
// C entry points of the CPU-only test library (libbbx_layout.so): build the
// layout of one orientation and run the emulator.  Not part of libbbx.so's ABI.
extern "C" {

// out[R] = A x for the R x C CSR matrix through the tiled layout + emulator
// (the G partial slabs are added in group order, like the epilogue kernels).
// info[0..8] = W, n_block, PR, G, n_quad, n_slice, n_extra, split_T, packed;
// packed: -1 = the builder's choice, 0 / 1 = plain ids / groups of five forced;
// gather_cycles = mean LDS cycles per ds_read_b64 half-wave group (or NULL).
int bbx_layout_emulate(int64_t R, int64_t C, int64_t nnz, const int32_t* rowptr,
                       const int32_t* colidx, const double* vals,
                       int bank_aware, int force_PR, int force_G,
                       int force_blocks, int max_threads, int chains,
                       int packed, const double* x, double* out,
                       int64_t* info, double* gather_cycles) {
  bbx::TiledOptions opt;
  opt.bank_aware = bank_aware != 0;
  opt.force_PR = force_PR;
  opt.force_G = force_G;
  opt.force_blocks = force_blocks;
  opt.stats = getenv("BBX_TILED_STATS") != nullptr;
  opt.chains = chains > 0 ? chains : 1;
  opt.packed = packed;
  if (max_threads > 0) opt.max_threads = max_threads;
  bbx::TiledHost m;
  std::string err;
  const std::vector<int64_t> rowptr64(rowptr, rowptr + R + 1);
  if (bbx::build_tiled_host(R, C, nnz, rowptr64.data(), colidx, vals, opt, &m,
                            &err) != 0) {
    fprintf(stderr, "bbx_layout_emulate: %s\n", err.c_str());
    return -1;
  }
  std::vector<double> slab;
  bbx::emulate_tiled_spmv(m, x, &slab);
  for (int64_t r = 0; r < R; ++r) {
    double a = 0.;
    for (int g = 0; g < m.G; ++g) a += slab[(size_t)g * (size_t)R + (size_t)r];
    out[r] = a;
  }
  if (info) {
    info[0] = m.W; info[1] = m.n_block; info[2] = m.PR; info[3] = m.G;
    info[4] = m.n_quad; info[5] = m.n_slice; info[6] = m.n_extra;
    info[7] = m.split_T; info[8] = m.packed ? 1 : 0;
  }
  if (gather_cycles) *gather_cycles = bbx::tiled_mean_gather_cycles(m);
  return 0;
}

// worker threads the builder would use now (affinity, cgroup quota, local ranks)
// the cost model's estimate (us) of one product in the layout for `chains`
double bbx_layout_model_cost(int64_t R, int64_t C, int64_t nnz,
                             const int32_t* rowptr, int chains) {
  if (!rowptr) return bbx::tiled_model_cost(R, C, nnz, nullptr, chains);
  const std::vector<int64_t> rowptr64(rowptr, rowptr + R + 1);
  return bbx::tiled_model_cost(R, C, nnz, rowptr64.data(), chains);
}

// X^T of a CSR with 64-bit index arrays (tests: against scipy's transpose).
// t_rowptr[C + 1], t_colidx[nnz], t_vals[nnz] or NULL.  Returns the structure
// check's bits (0 = transposed), -1 on a column id beyond int32.
int bbx_layout_transpose64(int64_t R, int64_t C, const int64_t* rowptr,
                           const int64_t* colidx, const double* vals,
                           int max_threads, int64_t* t_rowptr,
                           int32_t* t_colidx, double* t_vals) {
  const int64_t nnz = rowptr[R];
  const int bad = bbx::check_csr64_host(R, C, nnz, rowptr, colidx, max_threads);
  if (bad) return bad;
  std::vector<int32_t> narrow((size_t)nnz);
  for (int64_t k = 0; k < nnz; ++k) narrow[(size_t)k] = (int32_t)colidx[k];
  bbx::HostCsr t;
  bbx::transpose_csr_host(R, C, rowptr, narrow.data(), vals, max_threads, &t);
  memcpy(t_rowptr, t.rowptr.data(), sizeof(int64_t) * (size_t)(C + 1));
  if (nnz > 0) memcpy(t_colidx, t.colidx.data(), sizeof(int32_t) * (size_t)nnz);
  if (vals && t_vals && nnz > 0)
    memcpy(t_vals, t.vals.data(), sizeof(double) * (size_t)nnz);
  return 0;
}

int bbx_layout_builder_threads(int max_threads) {
  return bbx::builder_threads(max_threads > 0 ? max_threads : 64);
}

}  // extern "C"
#endif
