// The prior-preconditioned conjugate-gradient sampler, device resident.
//
// Replaces ConjugateGradientSampler.sample (reg_coef_sampler/cg_sampler.py:20-94)
// for precond_by='prior':
//   s, d                          cg_sampler.py:104,128-138
//   v = X~^T(sqrt(Omega) eta1) + phi eta2 ;  b = s (z + v)   :66-68
//   x = x0 / s                    :76
//   CG on  A x = d x + s X~^T Omega X~ (s x)                 :106-109
//   coef = s x                    :89
// and the SciPy >= 1.14 `cg` recurrence called at cg_sampler.py:77-80
// (scipy/sparse/linalg/_isolve/iterative.py `cg`, M = identity):
//   r = b - A x0
//   for k < maxiter: if ||r|| < atol: return; rho = r.r;
//                    p = r + (rho/rho_prev) p; q = A p; alpha = rho/(p.q);
//                    x += alpha p; r -= alpha q
//
// The loop's scalars live in a CGState on the device; the host only enqueues
// kernels and polls the `done` flag every few iterations.
#include <sys/prctl.h>
#include <time.h>

#include "common.hpp"
#include "philox.hpp"
#include "tiled_layout.hpp"

namespace bbx {

int launch_cg_setup(bbx_design* h, int n_unshrunk, const double* phi,
                    const double* sd, const double* x0, double* s, double* d,
                    double* xs, CGState* st, double atol);
int launch_cg_direction(bbx_design* h, int k, CGState* st,
                        const double* rr_part, const double* r, double* pvec,
                        const double* s, double* sp, double* c_part,
                        const double* d = nullptr, double* pdp_part = nullptr,
                        unsigned long long* word = nullptr,
                        unsigned long long tag = 0);
int launch_cg_update(bbx_design* h, int k, CGState* st, const double* pq_part,
                     const double* pvec, const double* q, double* x, double* r,
                     double* rr_part);
int launch_cg_finish(bbx_design* h, const double* s, const double* x,
                     double* coef);

__global__ __launch_bounds__(256) void fill_normal_kernel(
    int64_t len, uint64_t seed, uint64_t stream, double* __restrict__ out,
    int prio) {
  // (prio: the chain fills the NEXT draw's normals in front of its Polya-Gamma
  // kernel while the lambda kernel runs beside -- 49 us at normal priority, a
  // delay of the longer branch)
  if (prio) __builtin_amdgcn_s_setprio(3);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * blockDim.x) {
    Philox g(seed, stream, (uint64_t)i);
    out[i] = g.normal();
  }
}

__global__ __launch_bounds__(256) void any_nonzero_kernel(
    int64_t len, const double* __restrict__ x, int* __restrict__ flag) {
  bool nz = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
       i += (int64_t)gridDim.x * blockDim.x)
    nz = nz || (x[i] != 0.);  // NaN counts as non-zero, like ndarray.any()
  if (nz) atomicOr(flag, 1);
}

int launch_fill_normal(bbx_design* h, int64_t len, uint64_t seed,
                       uint64_t stream, double* d_out, bool prio,
                       hipStream_t on) {
  int64_t nb = (len + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  BBX_LAUNCH(fill_normal_kernel, dim3((unsigned)nb), dim3(256), 0,
                     on ? on : h->stream, len, seed, stream, d_out,
                     prio ? 1 : 0);
  BBX_HIP(hipGetLastError());
  return BBX_OK;
}

// One application of  q = d x + s X~^T (Omega (X~ (s x)))  given sp = s.*x and
// the partials of <offset, sp[1:]> already in PS_C.  Leaves partials of x.q in
// PS_PQ.
//
// With `upd` set (inside the CG loop, x == search direction p) the CG update of
// iteration upd->k rides in the Tdot epilogue whenever the format's dot kernel
// can deliver <t, Omega t>, t = X~ (s p):  p.Ap = <p, d p> + <t, Omega t>
// (algebraically SciPy's dotprod(p, q); a sum of non-negative terms), so alpha
// is known BEFORE q exists and q never has to be stored or re-read:
//   x += alpha p ; r -= alpha q ; partials of r.r -> PS_RR ; n_iter = k + 1.
// *upd->merged says whether that happened; if not, the caller runs
// cg_update_kernel on q as before.  One P-vector launch less per iteration.
struct CGUpdate {
  int k;
  CGState* st;
  double* x;
  double* r;
  bool* merged;
};
static int apply_operator(bbx_design* h, const double* d_omega,
                          const double* sp, const double* x, const double* s,
                          const double* d, double* q,
                          const CGUpdate* upd = nullptr) {
  TdotEpilogue ep;
  ep.mode = TD_OPER;
  ep.s = s;
  ep.d = d;
  ep.x = x;
  ep.dot_part = part_slot(h, PS_PQ);
  auto merge = [&]() {
    ep.mode = TD_OPER_UPD;
    ep.dot_part = part_slot(h, PS_RR);
    ep.cg_x = upd->x;
    ep.cg_r = upd->r;
    ep.cg_state = upd->st;
    ep.cg_k = upd->k;
    ep.pdp_part = part_slot(h, PS_PDP);
    ep.twt_part = part_slot(h, PS_TWT);
    *upd->merged = true;
  };
  if (upd) *upd->merged = false;
  // (dot and Tdot below belong to ONE application: t feeds the Tdot unchanged)
  struct OperatorScope {
    bbx_design* h;
    ~OperatorScope() { h->in_operator = false; }
  } op_scope{h};
  h->in_operator = true;
  h->operator_serial += 1;
  BBX_TRY(timer_begin(h, 2));  // family 2: the whole application (sampled)
  if (!h->sparse && dense_fused_applies(h)) {
    // f32 dense designs: both products in one pass over the matrix
    if (upd) merge();
    const int st = launch_operator_dense_fused(
        h, sp, d_omega, ep, q, upd ? part_slot(h, PS_TWT) : nullptr);
    if (st < 0) return st;
    if (st == 0) return timer_end(h, 2);
    return fail(BBX_ERR_STATE, "single-pass dense operator refused its design");
  }
  double* t = h->w_n[0].as<double>();
  int twt_done = 0;
  BBX_TRY(launch_dot(h, sp, d_omega, t, part_slot(h, PS_SUMW),
                     upd ? part_slot(h, PS_TWT) : nullptr, &twt_done));
  if (upd && twt_done) merge();
  BBX_TRY(launch_tdot(h, t, part_slot(h, PS_SUMW), ep, q));
  return timer_end(h, 2);
}

int cg_sample_device(bbx_design* h, const double* d_omega, const double* d_phi,
                     const double* d_z, const double* d_x0,
                     const double* d_sd, int n_unshrunk, const double* d_eta1,
                     const double* d_eta2, uint64_t seed, int maxiter,
                     double atol, double* d_coef, int* n_iter_out,
                     int* info_out, int x0_zero) {
  if (maxiter < 0) return fail(BBX_ERR_INVALID, "maxiter must be >= 0");
  if (n_unshrunk < 0 || n_unshrunk > h->P)
    return fail(BBX_ERR_INVALID, "n_unshrunk out of range");
  if ((d_eta1 == nullptr) != (d_eta2 == nullptr))
    return fail(BBX_ERR_INVALID,
                "randn_n and randn_P must both be given or both be NULL");
  double* s = h->w_P[0].as<double>();
  double* d = h->w_P[1].as<double>();
  double* x = h->w_P[2].as<double>();
  double* r = h->w_P[3].as<double>();
  double* pvec = h->w_P[4].as<double>();
  double* q = h->w_P[5].as<double>();
  // s.*p is the input of every X~ v inside the loop; the tiled kernel loads
  // v[intercept:] slice by slice with 16-byte accesses when that address is
  // 16-byte aligned, so the buffer starts one element in when there is an
  // intercept entry
  double* sp = h->w_P[6].as<double>() + (h->intercept ? 1 : 0);
  CGState* st = h->cg_state.as<CGState>();
  // Folded direction step (common.hpp DotFold; tiled format, one column group):
  // s.*r has its own buffer (same alignment rule as s.*p), the unscaled t of
  // the recurrence t_k = X~(s.*r_k) + beta t_{k-1} lives in the spare n-vector
  const bool fold = tiled_fold_applies(h);
  double* sr = h->w_P[9].as<double>() + (h->intercept ? 1 : 0);
  double* tu = h->w_n[2].as<double>();   // (eta1's buffer: consumed before the loop)

  if (d_eta1 == nullptr) {
    double* e1 = h->w_n[2].as<double>();
    double* e2 = h->w_P[8].as<double>();
    BBX_TRY(launch_fill_normal(h, h->n, seed, STREAM_ETA1, e1));
    BBX_TRY(launch_fill_normal(h, h->P, seed, STREAM_ETA2, e2));
    d_eta1 = e1;
    d_eta2 = e2;
  }

  BBX_TRY(launch_cg_setup(h, n_unshrunk, d_phi, d_sd, d_x0, s, d, x, st, atol));

  // Is the warm start all zeros?  SciPy's cg skips the product with x0 then
  // (`r = b - matvec(x) if x.any() else b.copy()`), and so do we.
  CGState* host_st = static_cast<CGState*>(h->host_pinned);
  if (x0_zero < 0) {
    int* d_flag = &st->pad;  // scratch word of the state cg_setup just reset
    int* h_flag = reinterpret_cast<int*>(host_st);
    BBX_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), h->stream));
    BBX_LAUNCH(any_nonzero_kernel, dim3(NPART), dim3(256), 0,
                       h->stream, h->P, d_x0, d_flag);
    BBX_HIP(hipGetLastError());
    BBX_HIP(hipMemcpyAsync(h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost,
                           h->stream));
    BBX_HIP(hipStreamSynchronize(h->stream));
    x0_zero = (*h_flag == 0) ? 1 : 0;
  }

  // r = b - A x0 with b = s (z + X~^T(sqrt(Omega) eta1) + phi eta2).
  // SciPy forms b by one transposed product and A x0 by an operator application
  // (two passes over X~^T for a warm start).  X~^T is linear, so both go
  // through ONE transposed product,
  //   g = X~^T (Omega (X~ (s x0)) - sqrt(Omega) eta1),
  //   r = s (z + (phi eta2 - g)) - d x0          (TD_RESID epilogue),
  // and b is never formed (only the absolute stop rule is in use).  One pass
  // over X~^T, one epilogue and one P-vector launch less per draw; the counters
  // then show one Tdot less than the reference's for a warm start.  (The
  // reference's sequence is kept as scripts/experiments/r02_cg_variants.patch.)
  {
    double* w = h->w_n[1].as<double>();
    TdotEpilogue ep;
    ep.mode = TD_RESID;
    ep.s = s;
    ep.d = d;
    ep.x = x0_zero ? nullptr : x;
    ep.z = d_z;
    ep.phi = d_phi;
    ep.eta2 = d_eta2;
    ep.dot_part = part_slot(h, PS_RR);
    if (fold) {   // the first X~ v kernel of the loop starts from s.*r
      ep.fold_sr = sr;
      ep.fold_cr_part = part_slot(h, PS_C);
    }
    if (!x0_zero && !h->sparse && dense_fused_applies(h)) {
      // single-pass dense operator: the normal term rides as a row addend
      BBX_TRY(launch_prep_v(h, x, s, sp, part_slot(h, PS_C)));
      BBX_TRY(launch_sqrt_scale(h, d_omega, d_eta1, w, part_slot(h, PS_SUMW),
                                nullptr, /*negate=*/true));
      const int st_f =
          launch_operator_dense_fused(h, sp, d_omega, ep, r, nullptr, w);
      if (st_f != 0)
        return st_f < 0 ? st_f
                        : fail(BBX_ERR_STATE,
                               "single-pass dense operator refused its design");
    } else {
      const double* t0 = nullptr;
      if (!x0_zero) {
        double* t = h->w_n[0].as<double>();
        BBX_TRY(launch_prep_v(h, x, s, sp, part_slot(h, PS_C)));
        BBX_TRY(launch_dot(h, sp, d_omega, t, nullptr));
        t0 = t;
      }
      BBX_TRY(launch_sqrt_scale(h, d_omega, d_eta1, w, part_slot(h, PS_SUMW),
                                t0, /*negate=*/t0 == nullptr));
      BBX_TRY(launch_tdot(h, w, part_slot(h, PS_SUMW), ep, r));
    }
  }

  // (the loop's CGState was reset by cg_setup_kernel: no upload, and no host
  // sync between the set-up and the loop -- the GPU used to idle ~30 us here
  // while the host woke up and refilled the queue)

  // ---- the loop.  The host only enqueues; what it knows of the solve it
  // reads from the PROGRESS WORD (common.hpp cg_word_store): the kernel that
  // carries the stop test of iteration k -- the direction kernel, or the X~ v
  // kernel of the folded iteration -- writes "k + 1 tests passed" or "done
  // after k iterations" into 8 bytes of coherent host memory, and the host
  // polls that address.  No read-back copy, no event, no stream
  // synchronisation inside a solve (rounds 2-5 looked at the flag through a
  // hipMemcpyAsync + event wait, first at (previous count + 2), then every
  // other iteration: an early look that failed left the GPU idle until the
  // host had woken up and refilled the queue, a late one cost four empty
  // launches per iteration enqueued in vain).
  //   * free run: iterations below the smallest count of the last solves are
  //     enqueued without looking (a solve rarely stops earlier: its kernels
  //     then return at entry, ~2 us each);
  //   * then iteration k is enqueued as soon as test k - AHEAD has passed: the
  //     queue always holds the rest of an iteration or more (no bubble), and
  //     when the rule fires at most AHEAD * launches-per-iteration empty
  //     launches sit between the stop and the finish kernel;
  //   * the finish kernel and the caller's tail (the chain's pass for X~ beta)
  //     go out the moment `done` is seen, and the call returns without waiting
  //     for them: `ev_poll` is recorded behind the finish kernel for whoever
  //     reads coef from another stream (coef_in_flight).
  // From here on the operator kernels look at CGState::done and exit at entry
  // once it is set.
  struct SkipScope {
    bbx_design* h;
    ~SkipScope() { h->skip_flag = nullptr; }
  } skip_scope{h};
  h->skip_flag = &st->done;
  double* pdp = part_slot(h, PS_PDP);
  bool merged = false;
  h->cg_serial += 1;
  const unsigned long long tag = (h->cg_serial & 0xFFFFFFull) << CG_WORD_TAG_SHIFT;
  unsigned long long* const word = h->cg_word_dev;
  volatile unsigned long long* const hword = h->cg_word_host;
  // one CG iteration after its direction kernel: q = A p and the update
  struct TagScope {
    bbx_design* h;
    ~TagScope() { h->timer.cur_tag = -1; }
  } tag_scope{h};
  auto operator_and_update = [&](int kk) -> int {
    h->timer.cur_tag = kk;  // kernel-timer samples know their iteration
    CGUpdate upd{kk, st, x, r, &merged};
    BBX_TRY(apply_operator(h, d_omega, sp, pvec, s, d, q, &upd));
    if (!merged)
      BBX_TRY(launch_cg_update(h, kk, st, part_slot(h, PS_PQ), pvec, q, x, r,
                               part_slot(h, PS_RR)));
    return BBX_OK;
  };
  // One CG iteration in THREE launches: the X~ v kernel carries the stop test
  // and the direction step (DotFold), the Tdot epilogue the update and s.*r.
  auto folded_iteration = [&](int kk) -> int {
    h->timer.cur_tag = kk;
    DotFold fa;
    fa.word = word;
    fa.tag = tag;
    fa.st = st;
    fa.k = kk;
    fa.intercept = h->intercept;
    fa.P = h->P;
    fa.rr_part = part_slot(h, PS_RR);
    fa.cr_part = part_slot(h, PS_C);
    fa.sr = sr;
    fa.tu = tu;
    fa.r = r;
    fa.pvec = pvec;
    fa.d = d;
    fa.pdp_part = pdp;
    BBX_TRY(timer_begin(h, 2));
    BBX_TRY(launch_dot_tiled_fold(h, fa, d_omega, h->w_n[0].as<double>(),
                                  part_slot(h, PS_SUMW), part_slot(h, PS_TWT)));
    TdotEpilogue ep;
    ep.mode = TD_OPER_UPD;
    ep.s = s;
    ep.d = d;
    ep.x = pvec;
    ep.dot_part = part_slot(h, PS_RR);
    ep.cg_x = x;
    ep.cg_r = r;
    ep.cg_state = st;
    ep.cg_k = kk;
    ep.pdp_part = pdp;
    ep.twt_part = part_slot(h, PS_TWT);
    ep.fold_sr = sr;
    ep.fold_cr_part = part_slot(h, PS_C);
    BBX_TRY(launch_tdot(h, h->w_n[0].as<double>(), part_slot(h, PS_SUMW), ep, q));
    return timer_end(h, 2);
  };
  auto iteration = [&](int kk) -> int {
    if (fold) return folded_iteration(kk);
    BBX_TRY(launch_cg_direction(h, kk, st, part_slot(h, PS_RR), r, pvec, s, sp,
                                part_slot(h, PS_C), d, pdp, word, tag));
    return operator_and_update(kk);
  };
  // how far ahead of the last passed test the host enqueues: one iteration
  // where an iteration outlasts the host's launches by far, more on small
  // designs (launch-bound: the host must not wait for every test)
  static const int ahead_env = getenv("BBX_CG_AHEAD") ? atoi(getenv("BBX_CG_AHEAD")) : 0;
  // (measured, profiles/r06_cg_word_ab.txt: 1M x 50k 103.3 / 104.6 / 105.0 us per
  // CG iteration with AHEAD = 1 / 2 / 3, the read-back look of round 5 104.3;
  // 100k x 10k 35.0 / 35.1 / 35.4, round 5 35.3)
  // small designs are launch-bound: 20k x 1k 29.5 / 29.1 / 29.9 with AHEAD = 1 / 2 /
  // 4 (round 5: 30.4), dense 6000 x 400 35.7 / 35.6 / 35.2 (r06_cg_word_small.txt)
  const int ahead = ahead_env >= 1 ? (ahead_env < 64 ? ahead_env : 64)
                    : h->n >= 50000 ? 1 : h->n >= 10000 ? 2 : 4;
  int free_run = 0;
  if (h->cg_recent_n > 0) {
    free_run = h->cg_recent[0];
    for (int i = 1; i < h->cg_recent_n; ++i)
      free_run = h->cg_recent[i] < free_run ? h->cg_recent[i] : free_run;
    free_run -= 1;   // (the solve that stops AT the smallest count is common)
  }
  if (free_run > maxiter) free_run = maxiter;
  // Between two tests the host has nothing to do.  It polls -- one core per
  // rank -- unless host cores are scarce: with fewer than three per rank
  // (affinity mask and cgroup quota over LOCAL_WORLD_SIZE: eight ranks on a
  // 16-CPU quota) and iterations longer than 60 us (1M x 50k: ~100 us) it
  // SLEEPS until 3/4 of the measured period after the last passed test and
  // polls from there; with AHEAD = 1 the queue then still holds the rest of
  // the running iteration (~85 us).  Measured on one rank at 1M x 50k
  // (profiles/r06_sleep_ab.txt): 0.87 instead of 1.68 busy cores, 105.7
  // instead of 104.4 us per CG iteration (late wake-ups) -- which is why it is
  // not the default where cores are plentiful.  BBX_CG_SLEEP=0 | 1 forces it.
  static const int sleep_env = getenv("BBX_CG_SLEEP") ? atoi(getenv("BBX_CG_SLEEP")) : -1;
  static const bool sleep_on = sleep_env >= 0 ? sleep_env != 0 : cores_per_rank() < 3;
  static thread_local bool slack_set = false;
  if (sleep_on && !slack_set) {
    (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);   // 1 us (default 50)
    slack_set = true;
  }
  auto now_ns = []() -> int64_t {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec;
  };
  *hword = tag;      // this solve: nothing passed yet (host store, before any launch)
  int k = 0;
  for (; k < free_run; ++k) BBX_TRY(iteration(k));
  bool done = false, bad = false;
  int n_iter = 0;
  unsigned spins = 0;
  int seen = 0;             // tests the host has seen pass
  int64_t t_seen = 0;       // ... and when it saw the last one
  bool slept = false;
  while (true) {
    const unsigned long long w = *hword;
    int passed = 0;
    if ((w >> CG_WORD_TAG_SHIFT) == (tag >> CG_WORD_TAG_SHIFT)) {
      if (w & CG_WORD_DONE) {
        done = true;
        bad = (w & CG_WORD_BAD) != 0;
        n_iter = (int)(w & 0xFFFFFFFFull);
        break;
      }
      passed = (int)(w & 0xFFFFFFFFull);
    }
    if (passed > seen) {
      const int64_t t = now_ns();
      if (t_seen && passed - seen <= 2) {
        // running mean of the period between two tests (this design)
        const int64_t per = (t - t_seen) / (passed - seen);
        h->cg_period_ns = h->cg_period_ns ? (3 * h->cg_period_ns + per) / 4 : per;
      }
      seen = passed;
      t_seen = t;
      slept = false;
    }
    if (k >= maxiter) break;       // exhausted: SciPy has no test after the last
    if (k < passed + ahead) {
      BBX_TRY(iteration(k));
      ++k;
      spins = 0;
      continue;
    }
    if (sleep_on && !slept && t_seen && ahead == 1 && h->cg_period_ns > 60000) {
      const int64_t until = t_seen + (h->cg_period_ns * 3) / 4;
      if (until - now_ns() > 15000) {
        timespec ts;
        ts.tv_sec = until / 1000000000LL;
        ts.tv_nsec = until % 1000000000LL;
        (void)clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &ts, nullptr);
        h->cg_naps += 1;
      }
      slept = true;     // (once per test: then poll)
      continue;
    }
    if (++spins > (1u << 14)) {
      // (a wedged queue shows up as a HIP error here instead of a host that
      // spins for ever; costs one driver call per ~16k polls)
      const hipError_t qe = hipStreamQuery(h->stream);
      if (qe != hipSuccess && qe != hipErrorNotReady) BBX_HIP(qe);
      spins = 0;
    }
    __builtin_ia32_pause();
  }
  h->timer.cur_tag = -1;
  h->tail_ran = false;
  h->coef_in_flight = false;
  BBX_TRY(launch_cg_finish(h, s, x, d_coef));
  if (done) {
    if (!h->ev_poll)
      BBX_HIP(hipEventCreateWithFlags(&h->ev_poll, hipEventDisableTiming));
    BBX_HIP(hipEventRecord(h->ev_poll, h->stream));
    h->coef_in_flight = true;
    if (h->tail_hook) {
      // (not timed, like the rest of the chain's kernels)
      const bool timing = h->timer.enabled;
      h->timer.enabled = false;
      h->skip_flag = nullptr;
      const int rc = h->tail_hook(h->tail_ctx);
      h->timer.enabled = timing;
      if (rc < 0) return rc;
      h->tail_ran = true;
    }
  } else {
    // maxiter exhausted (rare: a warning to the user): the device state says
    // how it ended
    BBX_HIP(hipMemcpyAsync(host_st, st, sizeof(CGState), hipMemcpyDeviceToHost,
                           h->stream));
    BBX_HIP(hipStreamSynchronize(h->stream));
    // (the test of the last enqueued iteration may have fired after the
    // host's last look)
    done = host_st->done != 0;
    bad = host_st->bad != 0;
    n_iter = host_st->n_iter;
  }
  // launches enqueued in vain: iterations n_iter .. k-1 minus the kernel that
  // ran the firing test
  if (k > n_iter) h->cg_empty_launches += (int64_t)(k - n_iter) * (fold ? 3 : 4) - 1;
  h->cg_solves += 1;
  timer_drop_skipped(h, n_iter);
  // Operator applications enqueued past the stopping iteration exited at entry
  // (their kernels see `done`): they are not matvecs and do not count
  // (abstract_matrix.py:61-72 counts products that ran).
  if (k > n_iter) {
    h->n_dot -= (k - n_iter);
    h->n_tdot -= (k - n_iter);
  }
  int info = done ? 0 : maxiter;
  if (bad) info = -1;
  h->last_cg_iter = n_iter;
  if (h->cg_recent_n < 8) {
    h->cg_recent[h->cg_recent_n++] = n_iter;
  } else {
    for (int i = 0; i < 7; ++i) h->cg_recent[i] = h->cg_recent[i + 1];
    h->cg_recent[7] = n_iter;
  }
  if (n_iter_out) *n_iter_out = n_iter;
  if (info_out) *info_out = info;
  if (info < 0)
    return fail(BBX_ERR_NUMERIC, "non-finite residual inside CG");
  return info;
}

}  // namespace bbx
