/*
 * bbx.h -- C ABI of libbbx.so: the MI355X (gfx950) implementation of the
 * bayes-bridge CG-accelerated regression-coefficient sampler.
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes
 * plain pointers and sizes, returns an int status and never throws.  The
 * reference interface each one replaces is cited as file:line relative to the
 * upstream repository (OHDSI/bayes-bridge, version 0.2.6).
 *
 * Status convention (mirrors SciPy's `info` of scipy.sparse.linalg.cg, which
 * the reference inspects at cg_sampler.py:82-92):
 *     0   success
 *   < 0   invalid input / runtime failure (see bbx_last_error())
 *   > 0   only from the CG entry points: the solver stopped at `maxiter`
 *         without reaching the tolerance; the result is still written, as the
 *         reference does (it warns and continues, cg_sampler.py:82-87).
 *
 * Threading/ownership: a handle is bound to one HIP device and one stream and
 * is NOT thread-safe (the reference is single-threaded and uses the process
 * global RNG, cg_sampler.py:51-62).  Host pointers are caller-owned and are
 * only read/written for the duration of the call.  The library owns the device
 * copies of X (both orientations) and all persistent work vectors.
 *
 * The ctypes precedent in the reference for such a boundary is
 * design_matrix/mkl_matvec.py:17-56 (MKL `mkl_dcsrmv`).
 */
#ifndef BBX_H
#define BBX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 103: + bbx_design_cg_stats (solves and launches enqueued past the stopping
 *      iteration since the last reset), bbx_launch_count, bbx_chain_set_progress.
 * 102: + bbx_design_create_csr64 (64-bit index arrays, 2^31 or more entries).
 * 101: bbx_design_tiled_info takes nine pointers (`packed`, since round 4),
 *      bbx_setup_lock_acquire/_release, bbx_design_useful_bytes.  A binding
 *      compares bbx_version() with the BBX_VERSION it was written against
 *      (bayesbridge_amd/_lib.py does) instead of calling with a stale arity. */
#define BBX_VERSION 103 /* 0.1.3 */

/* status codes */
#define BBX_OK 0
#define BBX_ERR_INVALID (-1)   /* bad argument (shape, NULL pointer, ...)      */
#define BBX_ERR_HIP (-2)       /* a HIP runtime call failed                     */
#define BBX_ERR_NODEVICE (-3)  /* no usable gfx950 device                       */
#define BBX_ERR_NUMERIC (-4)   /* NaN/Inf or non-positive curvature inside CG   */
#define BBX_ERR_STATE (-5)     /* call order violated (e.g. chain not set up)   */

/* storage formats of the sparse operator (bbx_design_create_csr `format`) */
#define BBX_FORMAT_AUTO 0  /* tiled when it applies, else csr                   */
#define BBX_FORMAT_CSR 1   /* reference layout: f64 values + i32 indices, CSR of
                              X and CSR of X^T (sparse_matrix.py:49,96,126)     */
#define BBX_FORMAT_TILED 2 /* LDS-tiled panels: u16 block-local indices, values
                              dropped when every stored entry is 1.0 (the idea
                              of cython_matmal/binary_matmul.pyx:21-25)         */

/* dtypes of dense storage */
#define BBX_F64 0
#define BBX_F32 1

/* likelihood families of a chain (model/factory.py:53-66) */
#define BBX_MODEL_LINEAR 0
#define BBX_MODEL_LOGIT 1

/* global-scale update of a chain: SamplerOptions.gscale_update
 * (gibbs_util.py:9-31; bayesbridge.py:412-448) */
#define BBX_GSCALE_SAMPLE 0   /* 'sample': conjugate Gamma draw of tau^-alpha  */
#define BBX_GSCALE_OPTIMIZE 1 /* 'optimize': Monte-Carlo EM step
                                 (bayesbridge.py:450-456)                       */
#define BBX_GSCALE_FIXED 2    /* None: tau stays at its current value           */

typedef struct bbx_design bbx_design; /* opaque: one design operator on one GPU */
typedef struct bbx_chain bbx_chain;   /* opaque: one device-resident Gibbs chain */
typedef struct bbx_batch bbx_batch;   /* opaque: chains that share the passes over X */

/* ---------------------------------------------------------------- library */

int bbx_version(void);
/* Thread-local text of the last failure; never NULL. */
const char* bbx_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only box; never touches a GPU
 * context beyond counting). */
int bbx_device_count(int* count);
/* Host worker threads the layout builder of a sparse design uses in this
 * process: the affinity mask, capped by the cgroup CPU quota (a box may show
 * 256 cores and grant 16), shared evenly among the LOCAL_WORLD_SIZE ranks
 * torch.distributed.run started on this node, at most 64;
 * BBX_BUILD_THREADS=N overrides.  (The reference builds nothing on extra
 * threads: SciPy's CSR is used as is, sparse_matrix.py:21-49.) */
int bbx_builder_threads(int* count);
/* Ranks that SHARE a GPU take the device-heavy part of their set-up one at a
 * time: when the environment variable BBX_SETUP_LOCK names a lock file, the
 * constructors hold an exclusive flock on it around their device work, and a
 * host wrapper brackets its own device set-up (data generation) with this
 * pair.  The lock is process-global and RE-ENTRANT for the thread that holds
 * it: nested acquires -- a constructor called inside a bracket -- only count;
 * another thread of the same process waits until the holder's last release,
 * and a release from a thread that does not hold it is ignored.  acquire returns 1 when
 * the lock is held afterwards (release it), 0 when BBX_SETUP_LOCK is unset or
 * the file cannot be opened (a note goes to stderr; nothing to release).
 * (No counterpart in the reference: one chain per process, one process per
 * device, bayesbridge.py:109.) */
int bbx_setup_lock_acquire(void);
int bbx_setup_lock_release(void);

/* ---------------------------------------------------- design operator (L1) */

/*
 * Build the operator  X~ = [1_n | X - 1_n offset^T]  (never materialised) from
 * a host CSR matrix.  Replaces SparseDesignMatrix.__init__
 * (design_matrix/sparse_matrix.py:21-49) for the part after zero-variance
 * column removal (done by the host wrapper, abstract_matrix.py:93-107).
 *   indptr[n+1], indices[nnz]: int32, as SciPy CSR; data[nnz] f64 or NULL when
 *   every stored value is 1.0; col_offset[p] f64 or NULL (= zeros, i.e. not
 *   centred); add_intercept: 1 => shape is (n, p+1) (sparse_matrix.py:51-54).
 * The structure is validated on the device (both constructors): indptr must
 * run 0 .. nnz non-decreasing, column indices must lie in [0, p) and ascend
 * within a row (duplicates allowed, they add up like in SciPy's csr_matvec;
 * SciPy users call X.sort_indices() first) -- otherwise BBX_ERR_INVALID.
 */
int bbx_design_create_csr(int64_t n, int64_t p, int64_t nnz,
                          const int32_t* indptr, const int32_t* indices,
                          const double* data, const double* col_offset,
                          int add_intercept, int device, int format,
                          bbx_design** out);

/*
 * The same from 64-BIT index arrays: what scipy.sparse.csr_matrix holds once a
 * matrix has 2^31 or more stored entries (SparseDesignMatrix keeps whatever
 * SciPy built, design_matrix/sparse_matrix.py:49; scipy.sparse picks int64 by
 * get_index_dtype), or when it was assembled from int64 arrays.  n and p must
 * still fit int32 (column and row ids are stored as 32-bit or narrower).
 *   nnz < 2^31: narrowed copies go through bbx_design_create_csr -- same
 *     formats, same validation, same results bit for bit.
 *   nnz >= 2^31: validation, the all-ones test and the transposition run on
 *     the HOST (threads: bbx_builder_threads), the design is stored in the
 *     LDS-tiled layout only -- format must be BBX_FORMAT_AUTO or
 *     BBX_FORMAT_TILED (the reference-layout kernels index with int32), and
 *     the layouts for batched chains (bbx_batch_*) are refused with
 *     BBX_ERR_STATE: such a design runs one chain at a time.
 */
int bbx_design_create_csr64(int64_t n, int64_t p, int64_t nnz,
                            const int64_t* indptr, const int64_t* indices,
                            const double* data, const double* col_offset,
                            int add_intercept, int device, int format,
                            bbx_design** out);

/* Same, but indptr/indices/data/col_offset are DEVICE pointers on `device`
 * (used when the matrix is generated on the GPU; the arrays are copied, the
 * caller keeps ownership). */
int bbx_design_create_csr_dev(int64_t n, int64_t p, int64_t nnz,
                              const int32_t* d_indptr, const int32_t* d_indices,
                              const double* d_data, const double* d_col_offset,
                              int add_intercept, int device, int format,
                              bbx_design** out);

/*
 * Dense operator.  Replaces DenseDesignMatrix.__init__/dot/Tdot
 * (design_matrix/dense_matrix.py:9-27,37-52).  X is row-major n x p (C order,
 * as NumPy), WITHOUT the intercept column; centring (col_offset != NULL) and
 * the intercept column are applied while the device copy is made, so the host
 * array is not modified (the reference centres the caller's array in place,
 * dense_matrix.py:21-22).  in_dtype is the dtype of X, storage_dtype the dtype
 * kept in HBM (BBX_F32 halves the traffic; all arithmetic stays f64).
 */
int bbx_design_create_dense(int64_t n, int64_t p, const void* X, int in_dtype,
                            int storage_dtype, const double* col_offset,
                            int add_intercept, int device, bbx_design** out);
/* X is a DEVICE pointer (row-major n x p, in_dtype). */
int bbx_design_create_dense_dev(int64_t n, int64_t p, const void* d_X,
                                int in_dtype, int storage_dtype,
                                const double* d_col_offset, int add_intercept,
                                int device, bbx_design** out);

int bbx_design_destroy(bbx_design* h);

/* shape -> (n, P) with P = p + add_intercept (sparse_matrix.py:51-54);
 * nnz -> stored entries of X_main (sparse_matrix.py:60-66; n*p for dense). */
int bbx_design_shape(const bbx_design* h, int64_t* n, int64_t* P);
int bbx_design_nnz(const bbx_design* h, int64_t* nnz);
int bbx_design_is_sparse(const bbx_design* h, int* flag);
/* 1 for a sparse design whose stored entries all equal 1.0 (kept value-free,
 * the counterpart of cython_matmal/binary_matmul.pyx:21-25), else 0. */
int bbx_design_is_binary(const bbx_design* h, int* flag);
/* HIP device index the handle lives on (the reference's counterpart is the
 * CuPy array's device, sparse_matrix.py:35). */
int bbx_design_device(const bbx_design* h, int* device);
/* Format actually in use (BBX_FORMAT_CSR / BBX_FORMAT_TILED; 0 for dense). */
int bbx_design_format(const bbx_design* h, int* format);
/* HBM bytes held by the operator's matrix storage (both orientations; a dense
 * design's transposed copy exists once a batch of chains has run on it). */
int bbx_design_storage_bytes(const bbx_design* h, int64_t* bytes);
/* Algorithmic HBM bytes of one dot / one Tdot in the storage format actually
 * read (nnz*(b_val+b_idx) + row pointers + in + out). */
int bbx_design_matvec_bytes(const bbx_design* h, int64_t* dot_bytes,
                            int64_t* tdot_bytes);
/* The part of those bytes that the kernels TIMED by bbx_design_get_timing
 * (which = 0 / 1) move.  Differs from bbx_design_matvec_bytes where a product
 * is more than one kernel and only the main one is stamped: the tiled and
 * dense Tdot write partial slabs that a separate epilogue kernel adds up
 * (the epilogue's slab read and its P-vector output are not counted here). */
int bbx_design_timed_bytes(const bbx_design* h, int64_t* dot_bytes,
                           int64_t* tdot_bytes);
/* The USEFUL part of bbx_design_timed_bytes: the stored entries at the
 * layout's index rate (tiled: 2 bytes per entry, 1.6 in groups of five, + 8 per
 * stored value -- no padding of the 16-byte steps, no schedules), the slices'
 * row ids, the vector in, the output.  *pad_dot / *pad_tdot = the share of the
 * id (and value) stream of X / X^T that is padding (0 for layouts without).
 * SURVEY.md 8(d) credits a kernel with the bytes it moves; this is the figure
 * beside it that a format with less padding would also have to move.  Mixed
 * designs (bbx_design_hybrid_info) and the other layouts report the timed
 * bytes.  Any output pointer may be NULL. */
int bbx_design_useful_bytes(const bbx_design* h, int64_t* dot_bytes,
                            int64_t* tdot_bytes, double* pad_dot,
                            double* pad_tdot);
/* Kernel launches per CG iteration of bbx_cg_sample / the chains on this design
 * (the loop of scipy.sparse.linalg.cg called at cg_sampler.py:77-80): 3 where
 * the direction step rides in the X~ v kernel and the update in the X~^T w
 * epilogue (tiled value-free layout with one column group), 4 where only the
 * update is merged, 5 otherwise.  Inside the CG loop the X~ v kernel that
 * bbx_design_timed_bytes describes then also moves 8 P-vectors. */
int bbx_design_cg_launches(const bbx_design* h, int* per_iteration);
/* The host side of that loop (cg_sampler.py:77-80 is a Python loop around two
 * products; here the host only enqueues, runs ahead of the device's stop test
 * and reads its outcome from a host-mapped progress word): CG solves on this
 * design since creation / the last reset, and the kernel launches they enqueued
 * past their stopping iteration (such kernels return at entry, ~2 us each);
 * `naps`: how often the host slept between two stop tests instead of polling
 * (only with fewer than three host cores per rank, or BBX_CG_SLEEP=1).
 * reset != 0 zeroes all after reading.  Output pointers may be NULL. */
int bbx_design_cg_stats(bbx_design* h, int64_t* solves, int64_t* empty_launches,
                        int64_t* naps, int reset);
/* Kernel launches this process has made through the library, all designs and
 * chains (a diagnostic: launches per second and rank is what the host side of
 * an N-rank node has to sustain; the reference launches nothing). */
uint64_t bbx_launch_count(void);
/* The 3-launch form: the direction step inside the X~ v kernel -- every
 * workgroup re-adds the r.r partials (rho, stop test, beta), the kernel streams
 * X~ (s.*r) and its epilogue forms t_k = X~ (s.*r_k) + beta t_{k-1}, which is
 * X~ (s.*p_k) by linearity.  Costs the X~ v kernel two n-vector passes, saves a
 * P-vector launch: +3.5 % Gibbs it/s at 100k x 10k, -1 % at 1M x 50k, hence ON by
 * default for designs of up to 250 000 rows where it applies.  on = 1 / 0
 * switches it for this design, -1 restores the default (BBX_CG_FOLD=0|1 sets
 * it for the process).  Same recurrence as scipy.sparse.linalg.cg
 * (cg_sampler.py:77-80) either way, to rounding. */
int bbx_design_set_cg_fold(bbx_design* h, int on);
/* Algorithmic bytes of the single-pass dense operator kernel X^T(Omega (X v))
 * (dense designs that qualify for it; 0 otherwise): one pass over the stored
 * matrix + v + Omega + the per-workgroup slabs it writes. */
int bbx_design_fused_operator_bytes(const bbx_design* h, int64_t* bytes);

/* Mixed designs (binary covariates plus a few continuous ones: the reference's
 * tests build simulate_design(n, p, binary_frac=.9), tests/helper.py:13) are
 * stored split by VALUE in the tiled format: the entries equal to 1.0 in the
 * value-free layout, the other entries of columns that hold many of them in a
 * dense column-major f64 block, what is left in the valued layout; the three
 * products are added in a fixed order.  *is_hybrid says whether that happened,
 * the counts how the entries were divided. */
int bbx_design_hybrid_info(const bbx_design* h, int* is_hybrid,
                           int64_t* ones_nnz, int64_t* rest_nnz,
                           int64_t* dense_nnz, int* dense_cols);
/* Geometry of the tiled format (BBX_FORMAT_TILED only): which = 0 for X, 1 for
 * X^T; W = column-block width, n_block = column blocks, PR = rows per panel,
 * G = column-block groups (partial-sum slabs), n_quad = 1024-byte steps (16
 * bytes x 64 lanes: per lane four 16-bit ids, or -- *packed = 1, value-free
 * designs -- one group of up to five entries, for each of its two rows),
 * n_slice = 128-row slices.  Any output pointer may be NULL. */
int bbx_design_tiled_info(const bbx_design* h, int which, int* W,
                          int* n_block, int* PR, int* G, int64_t* n_quad,
                          int64_t* n_slice, int* packed);

/*
 * out[n] = X~ v,  v[P].  Replaces SparseDesignMatrix.dot / main_dot
 * (sparse_matrix.py:68-101) and DenseDesignMatrix.dot (dense_matrix.py:37-48):
 *   out = v[0] + X_main v[1:] - <offset, v[1:]>.
 * Host pointers; synchronous.
 */
int bbx_design_dot(bbx_design* h, const double* v, double* out);
/*
 * out[P] = X~^T w,  w[n].  Replaces SparseDesignMatrix.Tdot / main_Tdot
 * (sparse_matrix.py:103-129) and DenseDesignMatrix.Tdot (dense_matrix.py:50-52):
 *   out = [sum(w) ; X_main^T w - sum(w) * offset].
 */
int bbx_design_tdot(bbx_design* h, const double* w, double* out);
/* Device-pointer forms: asynchronous on the handle's stream, which is a
 * NON-BLOCKING stream (it does not wait for the legacy null stream).  The
 * caller's buffers must be complete when the call is made (synchronise the
 * stream that produced them, or launch the producer on bbx_design_stream()),
 * and d_out is ready after bbx_design_synchronize().  The constructors taking
 * device pointers synchronise the whole device once before they start. */
int bbx_design_dot_dev(bbx_design* h, const double* d_v, double* d_out);
int bbx_design_tdot_dev(bbx_design* h, const double* d_w, double* d_out);

/*
 * out[P] = X~^T (obs_prec .* (X~ v)): the data part of the CG operator
 * (the closure at cg_sampler.py:106-109 without the diagonal scalings; also the
 * Hessian-matvec of the likelihood, logistic_model.py:62-78), issued through
 * the same launches the CG loop uses -- for dense designs that qualify this is
 * the single-pass kernel, so the entry lets a test compare it with the two
 * separate products.  Host pointers, synchronous; `_dev`: device pointers,
 * asynchronous on the handle's stream.
 */
int bbx_design_gram_matvec(bbx_design* h, const double* obs_prec,
                           const double* v, double* out);
int bbx_design_gram_matvec_dev(bbx_design* h, const double* d_obs_prec,
                               const double* d_v, double* d_out);

/* The hipStream_t (as void*) every kernel of this handle is launched on, and a
 * blocking wait on it. */
int bbx_design_stream(bbx_design* h, void** stream);
int bbx_design_synchronize(bbx_design* h);

/* ------------------------------------------------------- CG sampler (L3) */

/*
 * One draw  beta ~ N(Sigma z, Sigma),  Sigma^-1 = X~^T diag(obs_prec) X~ +
 * diag(prior_prec_sqrt^2), by perturbation-optimisation + prior-preconditioned
 * conjugate gradient.  Replaces ConjugateGradientSampler.sample
 * (reg_coef_sampler/cg_sampler.py:20-94) with precond_by='prior'
 * (cg_sampler.py:128-138), including precondition_linear_system
 * (cg_sampler.py:96-113) and the SciPy >= 1.14 `cg` recurrence it calls
 * (cg_sampler.py:77-80).
 *
 *   obs_prec[n]          Omega
 *   prior_prec_sqrt[P]   phi = 1/prior_sd (0 for a flat prior)
 *   z[P]                 X~^T (Omega y)
 *   x0[P]                coef_cg_init (CG warm start, in beta coordinates)
 *   precond_sd[P]        coef_scaled_sd; only the first n_unshrunk entries are
 *                        used: s_j = 2*precond_sd[j] (cg_sampler.py:133-136)
 *   randn_n[n], randn_P[P]  the standard-normal draws eta1, eta2 of
 *                        cg_sampler.py:61-62 (the reference draws them from the
 *                        global NumPy RNG, n first).  Pass NULL for BOTH to
 *                        draw them on the device from Philox4x32-10 keyed by
 *                        `seed` (distribution parity only).
 *   maxiter, atol        as cg_sampler.py:22-23; the stop rule is
 *                        ||r||_2 < atol in preconditioned coordinates.
 *   coef_out[P]          the draw; n_iter_out = number of completed CG
 *                        iterations (the reference's callback count);
 *   info_out             SciPy-style info: 0 converged, maxiter if exhausted.
 * Return value: 0, or > 0 (= info) when not converged, or < 0 on error.
 */
int bbx_cg_sample(bbx_design* h, const double* obs_prec,
                  const double* prior_prec_sqrt, const double* z,
                  const double* x0, const double* precond_sd, int n_unshrunk,
                  const double* randn_n, const double* randn_P, uint64_t seed,
                  int maxiter, double atol, double* coef_out, int* n_iter_out,
                  int* info_out);
/* All array arguments are DEVICE pointers; n_iter_out/info_out stay host. */
int bbx_cg_sample_dev(bbx_design* h, const double* d_obs_prec,
                      const double* d_prior_prec_sqrt, const double* d_z,
                      const double* d_x0, const double* d_precond_sd,
                      int n_unshrunk, const double* d_randn_n,
                      const double* d_randn_P, uint64_t seed, int maxiter,
                      double atol, double* d_coef_out, int* n_iter_out,
                      int* info_out);

/* Counters of operator applications since creation / last reset, the
 * equivalent of AbstractDesignMatrix.get_dot_count / reset_matvec_count
 * (abstract_matrix.py:61-72).  Applications inside bbx_cg_sample count too:
 * n_iter products with X~ and with X~^T for the iterations, one product with
 * X~ for a non-zero warm start, and ONE product with X~^T for the initial
 * residual -- the reference's two (the right-hand side's and the one inside
 * A x0) are a single pass over X~^T here, by linearity.  A pass is a pass:
 * it counts once. */
int bbx_design_matvec_count(const bbx_design* h, int64_t* n_dot,
                            int64_t* n_tdot);
int bbx_design_reset_matvec_count(bbx_design* h);

/* ------------------------------------------------ kernel timing (profiling) */

/* enabled = 1: HIP events are recorded on the handle's stream around every
 * dot / Tdot kernel launch (also those inside the CG loop and the chain);
 * enabled = N > 1: around every N-th launch of each family only (an event pair
 * costs a few microseconds of stream time, ~10% of a Gibbs iteration when
 * every launch is timed); 0: off. */
int bbx_design_set_timing(bbx_design* h, int enabled);
/* Resolves all pending event pairs (synchronises the stream) and returns the
 * number of timed launches and their summed device time per kernel family:
 * which = 0 dot (X v; for dense designs inside the CG loop: the single-pass
 * operator kernel), 1 Tdot (X^T w, main kernel), 2 one whole application of
 * the CG operator (dot + Tdot + epilogue; bracketed by two record commands,
 * which adds ~2-3 us of dispatch to the interval).  Launches that returned at
 * entry because their CG solve had already stopped (the host enqueues ahead
 * of the stop test) are not executions and are left out: samples below half
 * the family's median are dropped. */
int bbx_design_get_timing(bbx_design* h, int which, int64_t* n_launch,
                          double* total_ms);
int bbx_design_reset_timing(bbx_design* h);

/* Attainable-HBM probe ("confirm with a device-to-device
 * copy/stream on the box"): streams a scratch buffer of `bytes` through a
 * read-only kernel and a copy kernel `reps` times each, 16 bytes per lane, and
 * returns the HIP-event rates.  read_gbps counts bytes read; copy_gbps counts
 * bytes read + written.  Diagnostic only; not on the sampling path. */
int bbx_hbm_probe(int device, int64_t bytes, int reps, double* read_gbps,
                  double* copy_gbps);

/* ------------------------------------------- device-resident Gibbs chain */

/*
 * A whole Gibbs iteration of BayesBridge.gibbs(coef_sampler_type='cg')
 * (bayesbridge.py:210-240) kept in HBM: beta | rest by the CG sampler above,
 * Omega | beta (Polya-Gamma for logit, bayesbridge.py:397-410; Gamma for the
 * linear model's precision), tau | beta (bayesbridge.py:412-448), lambda |
 * tau, beta (tilted stable, bayesbridge.py:458-478), log-posterior
 * (bayesbridge.py:480-511), and the running summaries that give the CG warm
 * start and preconditioner scale (reg_coef_posterior_summarizer.py:3-124).
 * Random numbers come from Philox4x32-10 keyed by (seed, iteration, element):
 * distribution parity with the reference, never stream parity.
 *
 *   model          BBX_MODEL_LINEAR or BBX_MODEL_LOGIT
 *   outcome[n]     y (linear) or n_success (logit)          host pointer
 *   n_trial[n]     logit only; NULL => ones                  host pointer
 *   sd_unshrunk[n_unshrunk]  prior sd of the unshrunk coefficients
 *                  (intercept first; +inf = flat prior; bayesbridge.py:26-32)
 *   bridge_exp, slab_size    prior.py:9-15
 *   gscale_shape0, gscale_rate0  Gamma prior on tau^-bridge_exp (prior.py:77-81)
 * The chain borrows `design` (which must outlive it) and its stream.
 */
int bbx_chain_create(bbx_design* design, int model, const double* outcome,
                     const double* n_trial, int n_unshrunk,
                     const double* sd_unshrunk, double bridge_exp,
                     double slab_size, double gscale_shape0,
                     double gscale_rate0, uint64_t seed, bbx_chain** out);
int bbx_chain_destroy(bbx_chain* c);

/* Markov-chain state, host pointers.  obs_prec has n entries for logit and 1
 * for linear; lscale has P - n_unshrunk entries; gscale is in the RAW
 * parametrisation the sampler runs in (prior.py:129-141).  NULL = leave. */
int bbx_chain_set_state(bbx_chain* c, const double* coef,
                        const double* obs_prec, const double* lscale,
                        const double* gscale);
int bbx_chain_get_state(bbx_chain* c, double* coef, double* obs_prec,
                        double* lscale, double* gscale);
/* Running summaries (checkpoint/resume: bayesbridge.py:253-275,
 * reg_coef_sampler.py:42-58). */
int bbx_chain_set_summary(bbx_chain* c, const double* mean,
                          const double* square, int64_t n_averaged);
int bbx_chain_get_summary(bbx_chain* c, double* mean, double* square,
                          int64_t* n_averaged);
/* Omega at its initial value for the current coef: Polya-Gamma mean for logit
 * (logistic_model.py:80-87), 1/mean(resid^2) for linear
 * (bayesbridge.py:355-370). */
int bbx_chain_init_obs_prec(bbx_chain* c);
/* Iterations done so far (keys the Philox streams; settable for resume). */
int bbx_chain_get_iteration(bbx_chain* c, int64_t* iteration);
int bbx_chain_set_iteration(bbx_chain* c, int64_t iteration);
/* The Philox key of the chain (bbx_chain_create's `seed`); settable so that a
 * handle can continue another run's streams (gibbs_resume,
 * bayesbridge.py:43-107 restores the generator state the same way). */
int bbx_chain_get_seed(bbx_chain* c, uint64_t* seed);
int bbx_chain_set_seed(bbx_chain* c, uint64_t seed);
/* How tau is updated each iteration: BBX_GSCALE_SAMPLE (default),
 * BBX_GSCALE_OPTIMIZE or BBX_GSCALE_FIXED (bayesbridge.py:412-448 `method`). */
int bbx_chain_set_gscale_update(bbx_chain* c, int mode);
/*
 * Regenerates, on the device, the standard normals the chain's CG draw
 * consumes at 0-based iteration `iteration` (cg_sampler.py:61-62: eta1[n] for
 * the likelihood part, eta2[P] for the prior part) and copies them to the
 * host.  Philox is counter-based, so this does not disturb the chain; it
 * exists so that a test can feed the very same perturbation to the CPU oracle
 * and compare the device chain's draw exactly.
 */
int bbx_chain_eta(bbx_chain* c, int64_t iteration, double* eta1, double* eta2);
/* Log-likelihood and log-posterior of the state left by the last iteration
 * (bayesbridge.py:480-511), host pointers, either may be NULL. */
int bbx_chain_get_logp(bbx_chain* c, double* loglik, double* logp);

/*
 * Runs n_iter Gibbs iterations; a sample is kept every `thin` iterations
 * after `n_burnin` (gibbs_util.py:164-189), n_sample = (n_iter-n_burnin)/thin.
 *   maxiter, atol   of the CG solve; atol <= 0 => 1e-5*sqrt(P)
 *                   (reg_coef_sampler.py:95), maxiter <= 0 => 500.
 *   d_coef[n_sample*P]       DEVICE buffer, sample-major (sample s at s*P), or NULL
 *   d_lscale[n_sample*(P-n_unshrunk)], d_obs_prec[n_sample*n (logit) |
 *                   n_sample (linear)]   DEVICE buffers or NULL
 *   gscale[n_sample], logp[n_sample], n_cg_iter[n_sample]   HOST buffers or NULL
 * Returns 0, or the number of iterations whose CG solve hit maxiter (> 0), or
 * < 0 on error.
 */
int bbx_chain_run(bbx_chain* c, int n_iter, int n_burnin, int thin,
                  int maxiter, double atol, double* d_coef, double* d_lscale,
                  double* d_obs_prec, double* gscale, double* logp,
                  double* n_cg_iter);
/* Progress of a run (BayesBridge.gibbs(n_status_update=...), gibbs_util.py:
 * 214-238 prints "<k> Gibbs iterations complete: ..."): fn(iteration, ctx) is
 * called on the calling thread every `every` iterations of bbx_chain_run[_host],
 * after the coefficient draw of that iteration has been confirmed; every = 0
 * or fn = NULL switches it off.  The callback must not call back into the chain. */
int bbx_chain_set_progress(bbx_chain* c, int every, void (*fn)(int, void*),
                           void* ctx);
/* Same with HOST sample buffers for coef/lscale/obs_prec (copied at the end). */
int bbx_chain_run_host(bbx_chain* c, int n_iter, int n_burnin, int thin,
                       int maxiter, double atol, double* coef, double* lscale,
                       double* obs_prec, double* gscale, double* logp,
                       double* n_cg_iter);

/* ------------------------------------------------------------ batched chains
 * The reference runs ONE chain per process (bayesbridge.py:109) and its hot
 * loop is the operator of cg_sampler.py:105-108: two passes over the design per
 * CG iteration.  The matrix stream does not depend on the chain, so several
 * chains on one GPU can share every pass: a batch steps its chains in lock step
 * and runs the products of the CG solves (and the linear predictor of the
 * Omega update) as K-column products over one read of the matrix.  Everything
 * else of an iteration is the chain's own code with the chain's own Philox
 * keys; a chain's samples do not depend on which chains it is batched with
 * (bit for bit), and differ from `bbx_chain_run` only by the rounding of the
 * differently blocked sums.
 *
 * `chains`: n_chain chains created with bbx_chain_create on `design` -- 2 or 4
 * for sparse designs in the tiled format (2 when values are stored), 2, 4, 8,
 * 16 or 32 for dense designs, f32 or f64 storage (there the K-column products run
 * on the matrix cores, v_mfma_f64_16x16x4_f64, 16 chains per B operand; the
 * first batch builds a transposed copy of the matrix, as large as the matrix).  The batch borrows them:
 * set/get their state through the bbx_chain_* calls between runs, destroy the
 * batch before its chains.  The first batch of a width builds the matching
 * layout of the design (host pass, ~1 s at 1M x 50k). */
int bbx_batch_create(bbx_design* design, int n_chain, bbx_chain* const* chains,
                     bbx_batch** out);
/* What the library's cost model expects of a batch of n_chain chains on this
 * design: aggregate chain throughput of the batch / of the same chains run one
 * at a time, at the level of the operator's products (sparse: the geometry
 * search's per-tile + per-entry estimate of the K-layout against the
 * single-chain layout, csrc/tiled_layout.cpp; dense: two stream- or
 * MFMA-bound passes per application against the single-pass kernel).  No
 * layout is built.  bbx_batch_create REFUSES (BBX_ERR_INVALID) a width priced
 * below 1.0 -- e.g. 4 chains on the 1M x 50k design (0.67 predicted, 0.975
 * measured), 2 chains on an f32 dense design; bbx_batch_create_opts with
 * BBX_BATCH_ALLOW_SLOW builds it anyway (parity tests, measurements).  The
 * reference has no counterpart: one chain per process (bayesbridge.py:109). */
#define BBX_BATCH_ALLOW_SLOW 1u
int bbx_batch_predict(bbx_design* design, int n_chain, double* speedup);
int bbx_batch_create_opts(bbx_design* design, int n_chain,
                          bbx_chain* const* chains, unsigned flags,
                          bbx_batch** out);
int bbx_batch_destroy(bbx_batch* b);
/*
 * n_iter Gibbs iterations of every chain (arguments as bbx_chain_run).
 *   d_coef[n_chain]          host array of DEVICE buffers [n_sample * P], one per
 *                            chain (entries or the array itself may be NULL)
 *   gscale, logp, n_cg_iter  HOST buffers [n_chain * n_sample], chain-major, or NULL
 * Returns the number of (chain, iteration) CG solves that hit maxiter, or < 0.
 */
int bbx_batch_run(bbx_batch* b, int n_iter, int n_burnin, int thin,
                  int maxiter, double atol, double* const* d_coef,
                  double* gscale, double* logp, double* n_cg_iter);
/* per_chain[n_chain]: how many CG solves of each chain hit maxiter in the last
 * bbx_batch_run[_host] (its return value is their sum; the reference warns per
 * solve, cg_sampler.py:82-87). */
int bbx_batch_unconverged(const bbx_batch* b, int* per_chain);
/* Same with a HOST coefficient buffer [n_chain * n_sample * P] (chain-major,
 * sample s of chain c at (c * n_sample + s) * P), copied at the end, or NULL. */
int bbx_batch_run_host(bbx_batch* b, int n_iter, int n_burnin, int thin,
                       int maxiter, double atol, double* coef, double* gscale,
                       double* logp, double* n_cg_iter);
/* The batched products on their own, host pointers, chain-major: v [n_chain][P]
 * -> out [n_chain][n] (X~ v_c) and w [n_chain][n] -> out [n_chain][P] (X~^T w_c)
 * through the kernels the batch's CG loop launches (abstract_matrix.py:61-72's
 * dot / Tdot, K at a time).  For the parity tests. */
int bbx_batch_dot(bbx_batch* b, const double* v, double* out);
int bbx_batch_tdot(bbx_batch* b, const double* w, double* out);
/* Bytes ONE batched launch of each product kernel moves (all chains together):
 * the figure the kernel timers of the design are divided into for a batch. */
int bbx_batch_bytes(const bbx_batch* b, int64_t* dot_bytes,
                    int64_t* tdot_bytes);

/* The device-side scalar samplers on n_draw inputs (host pointers), exposed
 * so that their distributions can be tested against the host samplers:
 * Polya-Gamma(shape_i, tilt_i) and tilted stable(char_exp, tilt_i). */
int bbx_device_polya_gamma(int device, uint64_t seed, int64_t n_draw,
                           const int32_t* shape, const double* tilt,
                           double* out);
int bbx_device_tilted_stable(int device, uint64_t seed, int64_t n_draw,
                             double char_exp, const double* tilt, double* out);
int bbx_device_gamma(int device, uint64_t seed, int64_t n_draw, double shape,
                     double* out);
/* n_draw standard normals of Philox stream `stream` (element i = counter i):
 * the generator behind eta1 / eta2 (cg_sampler.py:61-62 uses
 * np.random.randn). */
int bbx_device_normal(int device, uint64_t seed, uint64_t stream,
                      int64_t n_draw, double* out);

/* ----------------------- host-side reference-stream samplers (libbbx_hostrng)
 * Exported by the separate, HIP-free libbbx_hostrng.so.  `bitgen` is the
 * address of a NumPy bitgen_t (PCG64(seed).ctypes.bit_generator); the draws
 * consume it exactly as random/polya_gamma/polya_gamma.pyx:40-74 and
 * random/tilted_stable/tilted_stable.pyx:65-134 do. */
int bbx_host_polya_gamma(void* bitgen, int64_t n, const int32_t* shape,
                         const double* tilt, double* out);
int bbx_host_tilted_stable(void* bitgen, int64_t n, const double* char_exp,
                           const double* tilt, double* out);
/* Checks of the device chain's Polya-Gamma arithmetic against the
 * reference-following one, on the host (both are in csrc/samplers.hpp):
 *   right_mass: log_form[i] = the mixture weight of the exponential piece at
 *     z[i] as polya_gamma.pyx:115-128 forms it (sums of logarithms),
 *     direct[i] = the device kernel's product form;
 *   series_accept: the alternating-series test (polya_gamma.pyx:139-162) of the
 *     proposal x[i] with the uniform u[i], 1 = accepted: sequential[i] with the
 *     terms of polya_gamma.pyx:131-137, direct[i] with the kernel's. */
int bbx_host_pg_right_mass(int64_t n, const double* z, double* log_form,
                           double* direct);
int bbx_host_pg_series_accept(int64_t n, const double* x, const double* u,
                              int32_t* sequential, int32_t* direct);

#ifdef __cplusplus
}
#endif
#endif /* BBX_H */
