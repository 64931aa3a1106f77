// Internal declarations shared by the translation units of libbbx.so.
// gfx950 (MI355X, CDNA4) only: 64-lane wavefronts are hard-coded.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/bbx.h"

// Every kernel launch of the library goes through these two and is counted
// (bbx_launch_count: bench.py reports launches per second and rank -- what an
// 8-rank node asks of its host cores).
#include <atomic>
namespace bbx {
extern std::atomic<unsigned long long> g_launch_count;
}
#define BBX_LAUNCH(...)                                                   \
  do {                                                                    \
    bbx::g_launch_count.fetch_add(1, std::memory_order_relaxed);          \
    hipLaunchKernelGGL(__VA_ARGS__);                                      \
  } while (0)
#define BBX_LAUNCH_EXT(...)                                               \
  do {                                                                    \
    bbx::g_launch_count.fetch_add(1, std::memory_order_relaxed);          \
    hipExtLaunchKernelGGL(__VA_ARGS__);                                   \
  } while (0)

namespace bbx {

constexpr int WAVE = 64;
// Grid of the P-length vector kernels and number of partial sums every
// two-stage reduction leaves behind.  Consumers re-add the NPART partials in a
// fixed order, so every reduction is bitwise reproducible run to run.
constexpr int NPART = 256;
constexpr int VEC_BLOCK = 256;

#if defined(__HIPCC__)
// Wave-level sums WITHOUT LDS traffic.  __shfl_down on a double is two
// ds_bpermute_b32 per step: a 64-lane tree costs 12 LDS round trips (~1.3k
// cycles of pure latency), which is most of what a latency-bound P-vector
// kernel or the row-sum exchange of the dense single-pass kernel waits for.
// Here: DPP moves inside a row of 16 lanes (v_mov_b32 row_ror:8/4/2/1), then
// v_readlane across the four rows.  Every lane returns the same value (the
// rotations pair the same operands in every lane and IEEE + commutes); fixed
// order, so bitwise reproducible like the tree it replaces.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// every lane of a row of 16 lanes gets the sum over its row
__device__ __forceinline__ double row16_allsum(double m) {
  m += dpp_move<0x128>(m);  // row_ror:8
  m += dpp_move<0x124>(m);  // row_ror:4
  m += dpp_move<0x122>(m);  // row_ror:2
  m += dpp_move<0x121>(m);  // row_ror:1
  return m;
}
__device__ __forceinline__ double lane_value(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, the same value in every lane
__device__ __forceinline__ double wave_allsum(double x) {
  const double m = row16_allsum(x);
  return (lane_value(m, 0) + lane_value(m, 16)) +
         (lane_value(m, 32) + lane_value(m, 48));
}
#endif

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

// The C ABI promises "no exceptions across the boundary": host-side set-up code
// (std::vector, std::thread) runs under this guard.
template <class F>
int no_throw(F&& f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    try {
      return fail(BBX_ERR_INVALID, "out of host memory");
    } catch (...) {
      return BBX_ERR_INVALID;
    }
  } catch (const std::exception& e) {
    try {
      return fail(BBX_ERR_INVALID, std::string("C++ exception: ") + e.what());
    } catch (...) {
      return BBX_ERR_INVALID;
    }
  } catch (...) {
    return BBX_ERR_INVALID;
  }
}

#define BBX_HIP(expr)                                                          \
  do {                                                                         \
    hipError_t err__ = (expr);                                                 \
    if (err__ != hipSuccess) {                                                 \
      return ::bbx::fail(BBX_ERR_HIP, std::string(#expr) + ": " +              \
                                          hipGetErrorString(err__));           \
    }                                                                          \
  } while (0)

#define BBX_TRY(expr)                                                          \
  do {                                                                         \
    int st__ = (expr);                                                         \
    if (st__ < 0) return st__;                                                 \
  } while (0)

// Chains that may share one pass over a sparse design (batched chains); the
// per-chain device pointers the batched kernels take by value.
constexpr int BATCH_MAX = 32;  // (sparse designs: 4; dense designs: 32)
// Dense batches interleave their vectors with a FIXED stride of 16 -- the 16
// columns of the MFMA's B operand -- whatever the number of chains; unused
// columns (and the rows of padding the kernels' last stages read) stay zero,
// so the operand loads of the product kernels carry no condition.
constexpr int DENSE_BATCH_STRIDE = 16;
constexpr int DENSE_BATCH_ROW_PAD = 320;  // zero rows past P / n of the 16-column operands
constexpr int DENSE_PAD_ROWS = 256;       // zero rows past n of the stored dense matrix
struct ChainPtrs {
  const double* p[BATCH_MAX];
};
struct ChainOut {
  double* p[BATCH_MAX];
};

// Device allocation owned by a handle; freed in the handle's destructor.
struct DevMem {
  void* ptr = nullptr;
  size_t bytes = 0;
  DevMem() = default;
  DevMem(const DevMem&) = delete;
  DevMem& operator=(const DevMem&) = delete;
  ~DevMem() { release(); }
  int alloc(size_t nbytes);
  void release();
  template <typename T>
  T* as() const {
    return static_cast<T*>(ptr);
  }
};

// Scalars of one CG solve, resident on the device so that the loop never has
// to round-trip to the host.
struct CGState {
  double rho[2];     // r.r of iteration k in slot k&1
  double atol;       // stop when ||r||_2 < atol   (cg_sampler.py:75-80)
  double bnorm2;     // ||b||^2 (diagnostic)
  int n_iter;        // completed iterations == callback count (cg_sampler.py:71-72)
  int done;          // 1 once the stop rule fired; later kernels exit at entry
  int bad;           // 1 if a non-finite or non-positive curvature was seen
  int pad;
  // !done, for work enqueued BEHIND the stop test that must only run once the
  // rule has fired (the chain's pass for X~ beta, bbx_design::tail_hook): such
  // kernels take &running as their skip flag
  int running;
  int pad2;
};

struct KernelTimer {
  bool enabled = false;
  int period = 1;          // time every period-th launch of each family
  // families: 0 dot kernel(s), 1 Tdot main kernel(s), 2 one whole operator
  // application of the CG loop (dot + Tdot + epilogue)
  static constexpr int FAMILIES = 3;
  int64_t seen[FAMILIES] = {0, 0, 0};
  bool armed[FAMILIES] = {false, false, false};
  struct Pair {
    hipEvent_t a, b;
    int tag;  // CG iteration the launch belongs to, -1 outside a CG loop
  };
  std::vector<Pair> pending[FAMILIES];
  std::vector<Pair> pool;
  // Launches enqueued past the stopping iteration exit at entry (skip_flag):
  // they take a few microseconds and are not executions of the kernel.  The
  // CG loop tags every sampled launch with its iteration (cur_tag) and, once
  // the solve's iteration count is known, discards the samples tagged beyond
  // it (timer_drop_skipped) -- no statistical filter.
  int cur_tag = -1;
  std::vector<float> samples[FAMILIES];
};

}  // namespace bbx

// One design operator on one GPU.  Layout of everything in HBM is described
// in DESIGN.md, section "Data layout".
struct bbx_design {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t n = 0, p = 0, P = 0, nnz = 0;
  int intercept = 0;
  bool sparse = true;
  bool binary = false;   // every stored value is 1.0 => values never read
  bool centred = false;
  int format = BBX_FORMAT_CSR;
  int dense_dtype = BBX_F64;

  // --- reference layout (BBX_FORMAT_CSR): CSR of X and CSR of X^T
  bbx::DevMem indptr, indices, data;        // X      : n rows
  bbx::DevMem t_indptr, t_indices, t_data;  // X^T    : p rows
  // X^T rows are split into chunks of <= T_CHUNK stored entries so that the
  // skewed column counts (1 ... 0.5 n, simulate_data.py:100-117) load-balance.
  bbx::DevMem t_chunk_row, t_chunk_begin;   // per chunk: row id, first entry
  bbx::DevMem t_row_chunk_ptr;              // per row : first chunk id  (p+1)
  int64_t n_tchunk = 0;
  bbx::DevMem t_partial;                    // per chunk partial sum

  // --- dense layout
  bbx::DevMem dense;  // row-major n x dense_ld (intercept column included,
                      // centred, zero padded), f32 or f64
  bbx::DevMem dense_slab;  // Tdot partial sums [dense_chunks][dense_ld]
  bbx::DevMem dense_fused_slab;  // fused operator: [workgroups][dense_ld]
  bbx::DevMem dense_batch_slab;  // batched Tdot: [row chunks][dense_ld][K]
  bbx::DevMem dense_xt;          // batched dot: X^T, row-major [dense_ld + pad][dense_xt_ld] f32
  int64_t dense_xt_ld = 0;
  bool dense_batch_attr = false;  // LDS size attribute of the batch kernels set on this device
  int dense_fused_wgs = 0;
  int64_t dense_ld = 0;
  int dense_chunks = 1;

  // --- LDS-tiled layout (BBX_FORMAT_TILED): see spmv_tiled.hip
  void* tiled = nullptr;           // bbx::TiledPair*
  void* hybrid = nullptr;          // bbx::HybridParts*: mixed designs (instead of `tiled`)
  // layouts sized for 2 and 4 right-hand sides (batched chains), built on
  // first use from the CSR arrays above
  void* tiled_k[2] = {nullptr, nullptr};
  // const bbx::HostCsr* of X and X^T WHILE a design created from 64-bit index
  // arrays with 2^31 or more entries is being built (bbx_design_create_csr64):
  // the layout builders read these instead of fetching device arrays.  Not owned.
  const void* host_csr[2] = {nullptr, nullptr};
  // bbx_batch_predict's answers, per width slot (2, 4, 8, 16, 32); < 0: not
  // asked yet.  (The sparse model fetches two index arrays from the device.)
  double batch_speedup[5] = {-1., -1., -1., -1., -1.};

  bbx::DevMem offset;  // column means (p), zeros when not centred

  // --- persistent work vectors
  bbx::DevMem w_n[3];     // n-length: t, eta1*sqrt(omega), spare
  bbx::DevMem w_P[10];    // P-length CG vectors
  bbx::DevMem part;       // NPART-length partial-sum slots (several)
  bbx::DevMem cg_state;   // CGState
  bbx::DevMem stage_n, stage_P;  // staging for the host-pointer entry points
  void* host_pinned = nullptr;   // small pinned buffer for flag read-back

  int64_t n_dot = 0, n_tdot = 0;
  int last_cg_iter = 0;  // iterations of the previous solve (poll scheduling)
  // The CG loop's PROGRESS WORD: 8 bytes of host memory mapped into the device
  // (fine-grained), written by the kernel that carries the stop test of an
  // iteration (cg_word_pack below) and polled by the host -- no read-back copy,
  // no event, no stream synchronisation inside a solve (cg_sampler.hip).
  unsigned long long* cg_word_host = nullptr;
  unsigned long long* cg_word_dev = nullptr;
  unsigned long long cg_serial = 0;   // solves so far: the word's tag
  int cg_recent[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // iteration counts of the last solves
  int cg_recent_n = 0;
  int64_t cg_period_ns = 0;   // running mean of the time between two stop tests
  // the last solve returned with its finish kernel enqueued, not waited for:
  // `ev_poll` is recorded behind it (another stream that reads coef waits on it)
  bool coef_in_flight = false;
  // what the last solve enqueued past its stopping iteration (bench.py)
  int64_t cg_empty_launches = 0, cg_solves = 0, cg_naps = 0;
  // Set around the dot + Tdot of ONE operator application (apply_operator,
  // gram_matvec): the Tdot's input is the dot's scaled output, so a mixed
  // design's dense block can ride in the dot kernel's epilogue for both
  // products (spmv_tiled.hip DenseEpi) instead of three kernels of its own.
  // Work the caller wants enqueued right behind a look at the stop flag, before
  // the host waits: it runs iff the rule has fired (skip flag = &CGState::running;
  // cg_sampler.hip).  tail_ran: it was enqueued at the look that found `done`.
  // Second destination of the draw (cg_finish_kernel writes coef there too): a
  // chain's sample slot for a kept iteration.  Set around ONE solve.
  double* coef_copy = nullptr;
  int (*tail_hook)(void*) = nullptr;
  void* tail_ctx = nullptr;
  bool tail_ran = false;
  hipEvent_t ev_poll = nullptr;   // recorded behind the draw's finish kernel (the tail may still run)
  bool in_operator = false;
  // counts operator applications: what the dot kernel of ONE application leaves
  // for its Tdot (a mixed design's D^T t partials) is tagged with it, so that
  // nothing of an earlier application -- a re-used buffer address -- is consumed
  uint64_t operator_serial = 0;
  // direction step folded into the X~ v kernel (DotFold): -1 = the default
  // (by size; BBX_CG_FOLD=0|1 for the process), 0 / 1 = bbx_design_set_cg_fold
  int cg_fold = -1;
  // Set around the CG loop: device address of CGState::done.  The host
  // enqueues operator applications ahead of the stop test; once the rule has
  // fired the big kernels see the flag and exit at entry instead of streaming
  // the matrix for an iteration that does not exist.
  const int* skip_flag = nullptr;
  bbx::KernelTimer timer;
};

namespace bbx {

// Partial-sum slots inside bbx_design::part (each NPART doubles).
enum PartSlot {
  PS_C = 0,     // <offset, v[1:]> of the current dot input
  PS_SUMW = 1,  // sum of the current Tdot input
  PS_PQ = 2,    // p.q
  PS_RR = 3,    // r.r
  PS_MISC = 4,  // 4, 5, 6: scratch triples
  PS_ZERO = 7,  // never written: NPART zeros
  PS_PDP = 8,   // <p, d .* p> of the current search direction
  PS_TWT = 9,   // <t, Omega t>, t = X~ (s .* p): with PS_PDP the curvature p.Ap
  PS_COUNT = 10
};

inline double* part_slot(const bbx_design* h, int slot) {
  return h->part.as<double>() + (size_t)slot * NPART;
}

// ---- operator launches (spmv_csr.hip / dense.hip) --------------------------
// t[n] = rowscale ? rowscale .* (X~ v) : X~ v ;  v is a device P-vector.
// `c_part` must hold the NPART partials of <offset, v[1:]> (see launch_prep_v).
// If sum_part != nullptr the NPART partial sums of t are written there.
// If d_twt_part != nullptr and the format's dot kernel can produce them, the
// NPART partials of sum_i rowscale_i t_i^2 (t before scaling) are written there
// and *twt_done is set to 1; otherwise *twt_done is 0 and nothing is written.
int launch_dot(bbx_design* h, const double* d_v, const double* d_rowscale,
               double* d_t, double* d_sum_part, double* d_twt_part = nullptr,
               int* twt_done = nullptr);
// Modes of the Tdot epilogue.
enum TdotMode {
  TD_PLAIN = 0,  // out = g
  TD_OPER = 1,   // out = d .* x + s .* g            (cg_sampler.py:107-108)
  // q = d .* p + s .* g as TD_OPER, then the CG update in the same pass:
  //   alpha = rho / p.Ap ; x += alpha p ; r -= alpha q ; partials of r.r
  // with p.Ap = <p, d p> + <t, Omega t> taken from two partial-sum slots (q is
  // not stored).  SciPy's cg: `alpha = rho_cur / dotprod(p, q)`.
  TD_OPER_UPD = 3,
  // The initial residual of a CG draw in ONE transposed product: with
  //   g = X~^T (Omega (X~ (s x0)) - sqrt(Omega) eta1)   (linearity of X~^T)
  //   r = b - A x0 = s .* (z + (phi .* eta2 - g)) - d .* x0     (x == nullptr:
  // cold start, r = b), partials of r.r.  SciPy forms b and A x0 separately
  // (two products with X~^T); b itself is not needed: the stop rule is the
  // absolute one (cg_sampler.py:75-80).
  TD_RESID = 4
};
struct TdotEpilogue {
  int mode = TD_PLAIN;
  const double* s = nullptr;
  const double* d = nullptr;
  const double* x = nullptr;
  const double* z = nullptr;
  const double* phi = nullptr;
  const double* eta2 = nullptr;
  double* dot_part = nullptr;  // TD_OPER: partials of x.out
                               // TD_OPER_UPD: partials of the new r.r
  // TD_OPER_UPD only (x above is the search direction p):
  double* cg_x = nullptr;      // iterate, updated in place
  double* cg_r = nullptr;      // residual, updated in place
  CGState* cg_state = nullptr;
  int cg_k = 0;
  const double* pdp_part = nullptr;
  const double* twt_part = nullptr;
  // TD_OPER_UPD / TD_RESID feeding a folded direction step (DotFold): also
  // write s.*r and the NPART partials of <offset, (s.*r)[1:]>
  double* fold_sr = nullptr;
  double* fold_cr_part = nullptr;
};
// The direction step of CG iteration k folded into the tiled X~ v kernel
// (spmv_tiled.hip tiled_spmv_kernel<.., FOLD = true>; cg_sampler.hip).  Instead
// of a separate launch that forms p = r + beta p and s.*p (cg_direction_kernel),
// the Tdot epilogue of iteration k-1 also leaves s.*r and the partials of
// <offset, (s.*r)[1:]>, and EVERY workgroup of the X~ v kernel
//   * re-adds the NPART partials of r.r in the fixed order -> rho, the stop
//     test of SciPy's loop top, beta = rho / rho_prev;
//   * streams  u = X~ (s.*r_k)  -- its LDS slices hold s.*r, ONE vector that
//     does not depend on beta -- and forms in its epilogue
//         t_k = u + beta t_{k-1}     ( = X~ (s.*p_k): X~ is linear and
//                                        s.*p_k = s.*r_k + beta s.*p_{k-1} )
//     from the previous iteration's unscaled t (an n-vector, updated in place);
// and the workgroups, each for its own contiguous share of the P coordinates,
// write p_k = r_k + beta p_{k-1} and the partials of <p, d p>.  s.*p is never
// formed.  Three launches per CG iteration instead of four, no grid-wide
// reduction inside a launch.  (Round 4's first form filled the slices with
// s.*r + beta s.*p_old -- two vector reads per slice -- and lost 7 us in the
// X~ v kernel for the 6.6 us it saved: LABNOTES.md R4.1.)
// Progress word of a solve (bbx_design::cg_word_*): [tag:24 | bad | done | value:32],
// value = stop tests passed so far (the rule has not fired: iterations 0 .. value-1
// are under way) or, with `done`, the solve's iteration count.
constexpr unsigned long long CG_WORD_DONE = 1ull << 32;
constexpr unsigned long long CG_WORD_BAD = 1ull << 33;
constexpr int CG_WORD_TAG_SHIFT = 40;
__device__ inline void cg_word_store(unsigned long long* word,
                                     unsigned long long v) {
  // system scope: the host polls this address
  __hip_atomic_store(word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct DotFold {
  unsigned long long* word = nullptr;   // progress word (device address) and
  unsigned long long tag = 0;           // this solve's tag, already shifted
  CGState* st = nullptr;
  int k = 0;
  int intercept = 0;
  int64_t P = 0;
  const double* rr_part = nullptr;  // NPART partials of r.r
  const double* cr_part = nullptr;  // NPART partials of <offset, (s.*r)[1:]>
  const double* sr = nullptr;       // s.*r      (P; [intercept:] 16-byte aligned)
  double* tu = nullptr;             // unscaled t = X~ (s.*p), n; read for k > 0
  const double* r = nullptr;
  double* pvec = nullptr;           // p, updated in place
  const double* d = nullptr;
  double* pdp_part = nullptr;       // NPART partials of <p, d p>
};

// The dense block D (kd continuous columns, column-major [kd][n]) of a mixed
// design inside the value-free X~ v kernel's epilogue: t_r += sum_j D[j][r] v_j,
// and -- the output w = rowscale .* t being the input of the operator's
// transposed product -- part[workgroup][j] = sum over its rows of D[j][r] w_r.
// D is read once per operator application instead of twice, and the addend /
// dense-Tdot kernels (three launches) disappear from the CG iteration.
// Dense columns the epilogue handles.  Its D stream runs at the kernel's tail,
// after the id stream: measured against the separate kernels at 1M x 50k
// (profiles/r04_mixed.txt) one chain runs +11 % / +9 % / +3 % faster with 2 / 5 /
// 8 continuous columns and -1 % / -2.5 % slower with 12 / 16.
constexpr int DENSE_EPI_MAX = 8;
struct DenseEpi {
  const double* D = nullptr;
  const int32_t* cols = nullptr;   // column of X (without intercept) of each D row
  int kd = 0;
  int64_t n = 0;
  double* part = nullptr;          // [workgroups][kd]
};

// out[P] = epilogue([sum w ; X_main^T w - sum(w) offset]).  `sumw_part` holds
// the NPART partials of sum(w).
int launch_tdot(bbx_design* h, const double* d_w, const double* d_sumw_part,
                const TdotEpilogue& ep, double* d_out);

// ---- vector kernels (vecops.hip) -------------------------------------------
// v = s ? s .* x : x  (written to d_v unless d_v == x and s == nullptr), and the
// NPART partials of <offset, v[1:]> into c_part.
// (skip: a device flag; the kernel returns at entry while it is non-zero)
int launch_prep_v(bbx_design* h, const double* d_x, const double* d_s,
                  double* d_v, double* d_c_part, const int* d_skip = nullptr);
// partials of sum(w .* a) (a may be nullptr => sum(w)) over n entries.
int launch_sum_n(bbx_design* h, const double* d_w, int64_t len,
                 double* d_part);
// w[i] = sqrt(omega[i]) * eta[i], plus partials of sum(w).
// minus != nullptr: w[i] = minus[i] - sqrt(omega[i]) * eta[i] (TD_RESID's input)
int launch_sqrt_scale(bbx_design* h, const double* d_omega,
                      const double* d_eta, double* d_w, double* d_part,
                      const double* d_minus = nullptr, bool negate = false);

int design_alloc_work(bbx_design* h);
// Registry of live design handles.  Chains and batches borrow their design; a
// garbage-collected host language may finalise a design BEFORE the chains bound
// to it (Python's cycle collector runs __del__ in arbitrary order): their
// destroy calls must then not touch the design's device or stream.
void design_register(const bbx_design* h);
void design_unregister(const bbx_design* h);
bool design_alive(const bbx_design* h);
int build_transpose_csr(bbx_design* h);
int launch_dot_csr(bbx_design* h, const double* d_v, const double* d_rowscale,
                   double* d_t);
int launch_tdot_csr(bbx_design* h, const double* d_w,
                    const double* d_sumw_part, const TdotEpilogue& ep,
                    double* d_out);
int launch_tdot_finalize(bbx_design* h, const double* d_gfull, int n_slab,
                         const double* d_sumw_part, const TdotEpilogue& ep,
                         double* d_out);
int launch_tdot_finalize_dense(bbx_design* h, const TdotEpilogue& ep,
                               double* d_out, const double* d_slab = nullptr,
                               int n_slab = 0);
// Dense designs stored in f32 with <= 8192 padded columns: one pass over the
// matrix computes  out = epilogue(X^T (rowscale .* (X v)))  (dense.hip).
// Returns 1 when the fused path does not apply (caller runs the two passes).
int launch_operator_dense_fused(bbx_design* h, const double* d_v,
                                const double* d_rowscale,
                                const TdotEpilogue& ep, double* d_out,
                                double* d_twt_part = nullptr,
                                const double* d_addend = nullptr);
// (d_addend: n doubles added to rowscale .* (X v) before the transposed product)
bool dense_fused_applies(const bbx_design* h);
int build_tiled(bbx_design* h);
void destroy_tiled(bbx_design* h);
// Does the CG loop on this design fold its direction step into the X~ v kernel?
// Possible on the tiled value-free layout with one column group and one
// partial slot per panel (not a mixed design); by default on for designs of up
// to 250 000 rows, where it measures faster (LABNOTES.md, round 4); per design
// bbx_design_set_cg_fold, per process BBX_CG_FOLD=0|1.
bool tiled_fold_applies(const bbx_design* h);
// t = rowscale .* (X~ (s.*p_k)) with the direction step of iteration fa.k inside
// (see DotFold); partials of sum(t) and of <t, Omega t> as launch_dot.
int launch_dot_tiled_fold(bbx_design* h, const DotFold& fa,
                          const double* d_rowscale, double* d_t,
                          double* d_sum_part, double* d_twt_part);
bool tiled_batch_value_free(const bbx_design* h);  // batches of any width (else pairs only)
int ensure_tiled_k(bbx_design* h, int K);
// predicted throughput of a batch of K chains / K single chains (cost model)
int tiled_batch_predict(const bbx_design* h, int K, double* speedup);
// Per-chain arguments of a batched launch of the tiled kernels (K > 1).
struct TiledBatchArgs {
  ChainPtrs rowscale{};   // dot epilogue: Omega of each chain (entries may be null)
  ChainOut out{};         // dot epilogue: out.p[c][row * out_stride]
  int out_stride = 1;
  int part_stride = 0;    // doubles between the chains' NPART-blocks
};
// t_c = rowscale_c .* (X~ v_c) for the K chains of the interleaved [P][K] input
// `d_v` (c_part: K NPART-blocks of <offset, v_c[1:]>, part_stride apart).  When
// d_sum_part != nullptr it receives per chain the NPART partials of sum(t_c)
// and, twt_off doubles further, of <t_c, Omega_c t_c>.
int launch_dot_tiled_k(bbx_design* h, int K, const double* d_v,
                       const double* d_c_part, const TiledBatchArgs& ba,
                       double* d_sum_part, int twt_off);
// Main kernel of X~^T w_c for the interleaved [n][K] input: leaves G slabs
// [G][p][K] (returned through slab / G) for the batched epilogue kernel.
int launch_tdot_tiled_k(bbx_design* h, int K, const double* d_w,
                        const double** slab, int* G);
int tiled_batch_bytes(const bbx_design* h, int K, int64_t* dot_bytes,
                      int64_t* tdot_bytes);
// The same two products for dense designs (f32 or f64 storage) on the matrix cores, K <= 32
// chains (dense_batch.hip): t_c = rowscale_c .* (X v_c) with the partials of
// <t_c, Omega_c t_c> (d_twt_part[c * NPART + ...], may be null), and the slabs
// [G][ld][16] of X^T w_c.  d_v is [ld + 64][16] and d_w [n + 64][16], zero
// outside the chains' columns and past P / n rows (DENSE_BATCH_STRIDE).
bool dense_batch_applies(const bbx_design* h);
int dense_batch_stride(int K);  // 16, or 32 for more than 16 chains
int launch_dot_dense_k(bbx_design* h, int K, const double* d_v,
                       const TiledBatchArgs& ba, double* d_twt_part);
int launch_tdot_dense_k(bbx_design* h, int K, const double* d_w,
                        const double** slab, int* G);
int dense_batch_bytes(const bbx_design* h, int K, int64_t* dot_bytes,
                      int64_t* tdot_bytes);
int dense_batch_predict(const bbx_design* h, int K, double* speedup);
// timed_only: count what the timed kernel of each family moves (tiled Tdot:
// without the epilogue kernel's slab read and P-vector output)
int tiled_matvec_bytes(const bbx_design* h, int64_t* dot_bytes,
                       int64_t* tdot_bytes, bool timed_only = false);
int tiled_useful_bytes(const bbx_design* h, int64_t* dot_bytes,
                       int64_t* tdot_bytes, double* pad_dot, double* pad_tdot);
int64_t tiled_storage_bytes(const bbx_design* h);
// 1 when the design is stored split by value (HybridParts), else 0
int tiled_hybrid_info(const bbx_design* h, int64_t* ones_nnz,
                      int64_t* rest_nnz, int64_t* dense_nnz, int* kd);
int tiled_describe(const bbx_design* h, int which, int* W, int* n_block,
                   int* PR, int* G, int64_t* n_quad, int64_t* n_slice,
                   int* packed = nullptr);
// (prio: ask the wave scheduler for priority; on: the HIP stream, null = the design's)
int launch_fill_normal(bbx_design* h, int64_t len, uint64_t seed,
                       uint64_t stream, double* d_out, bool prio = false,
                       hipStream_t on = nullptr);

// Ranks that share a GPU take the device part of a design's set-up one at a
// time: RAII form of bbx_setup_lock_acquire / _release (api.hip; a no-op unless
// BBX_SETUP_LOCK names a lock file; re-entrant inside a process).
struct SetupTurn {
  int held;
  SetupTurn() : held(bbx_setup_lock_acquire()) {}
  void release() {
    if (held > 0) bbx_setup_lock_release();
    held = 0;
  }
  ~SetupTurn() { release(); }
  SetupTurn(const SetupTurn&) = delete;
  SetupTurn& operator=(const SetupTurn&) = delete;
};

int timer_begin(bbx_design* h, int which);
int timer_end(bbx_design* h, int which);
// Single-kernel families: hands out the event pair of this launch (nullptr,
// nullptr when the launch is not sampled) to be passed to
// hipExtLaunchKernelGGL, which stamps the KERNEL's own begin and end (what
// rocprofv3 reports) instead of bracketing the dispatch with two record
// commands (~3 us more per launch).
int timer_arm(bbx_design* h, int which, hipEvent_t* a, hipEvent_t* b);
// Discards the pending samples of launches tagged with an iteration >= n_iter
// (they found the stop flag set and returned at entry).  The stream is idle.
void timer_drop_skipped(bbx_design* h, int n_iter);

// ---- CG sampler (cg_sampler.hip) -------------------------------------------
// POSTCONDITION: on return d_coef is final IN THE ORDER OF h->stream -- the
// finish kernel is enqueued, not waited for, and the stream may also hold the
// caller's tail (read-only with respect to coef).  When the stop rule fired
// (the normal case) h->coef_in_flight is set and h->ev_poll is recorded right
// behind the finish kernel: work on ANOTHER stream that reads d_coef must wait
// on that event (chain_post_draw does).  When maxiter was exhausted the solve
// ends with hipStreamSynchronize.  n_iter_out / info_out are final on return.
int cg_sample_device(bbx_design* h, const double* d_omega, const double* d_phi,
                     const double* d_z, const double* d_x0,
                     const double* d_sd, int n_unshrunk, const double* d_eta1,
                     const double* d_eta2, uint64_t seed, int maxiter,
                     double atol, double* d_coef, int* n_iter_out,
                     int* info_out, int x0_zero = -1);
// x0_zero: 1 the warm start is known to be all zeros (the initial-residual
// product is skipped, as SciPy does), 0 known not to be, -1 check on the device.

}  // namespace bbx
