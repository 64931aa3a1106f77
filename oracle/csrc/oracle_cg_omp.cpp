// Multi-core CPU baseline of the CG coefficient sampler (TEST INFRASTRUCTURE:
// the `cpu_baseline.port_omp` leg of bench.py and its own parity test; never
// linked into or called by the product).
//
// Same algorithm as oracle/cg_sampler.py (which restates
// reg_coef_sampler/cg_sampler.py:20-151 and SciPy's `cg`), written as plain
// C++ loops with OpenMP over rows:
//   X~ v   = v0 + X v1 - <offset, v1>      one thread-parallel pass over CSR(X)
//   X~^T w = [sum w ; X^T w - sum(w) off]  one pass over CSR(X^T) (built by the
//                                          caller with scipy), so no atomics
// Reductions are OpenMP sums: deterministic for a fixed thread count only,
// which is why this is a *baseline* and the single-thread NumPy oracle stays
// the parity checker.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <omp.h>

namespace {

// One orientation in CSR, owned by the library: the arrays are allocated
// untouched and FIRST TOUCHED by the threads that will stream them (contiguous
// row ranges of equal stored entries), so that on a multi-socket host every
// thread reads its share from its own NUMA node.  NumPy/SciPy arrays handed in
// from Python are touched by one thread and sit on one node: with them the
// products ran at ~50 GB/s on a 2-socket box whatever the thread count.
struct Csr {
  int64_t rows = 0, nnz = 0;
  int32_t* indptr = nullptr;
  int32_t* indices = nullptr;
  double* data = nullptr;  // nullptr: every stored value is 1.0
  std::vector<int64_t> bound;  // thread t owns rows [bound[t], bound[t + 1])

  void release() {
    free(indptr);
    free(indices);
    free(data);
    indptr = indices = nullptr;
    data = nullptr;
  }
  bool adopt(int64_t n_rows, const int32_t* ip, const int32_t* ix,
             const double* da, int n_threads) {
    rows = n_rows;
    nnz = ip[n_rows];
    indptr = static_cast<int32_t*>(malloc(sizeof(int32_t) * (size_t)(rows + 1)));
    indices = static_cast<int32_t*>(malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1)));
    if (da) data = static_cast<double*>(malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1)));
    if (!indptr || !indices || (da && !data)) return false;
    bound.assign((size_t)n_threads + 1, rows);
    bound[0] = 0;
    for (int t = 1; t < n_threads; ++t) {
      const int64_t want = nnz / n_threads * t;
      bound[(size_t)t] = std::lower_bound(ip, ip + rows, (int32_t)want) - ip;
      if (bound[(size_t)t] < bound[(size_t)t - 1]) bound[(size_t)t] = bound[(size_t)t - 1];
    }
#pragma omp parallel num_threads(n_threads)
    {
      const int t = omp_get_thread_num();
      const int64_t r0 = bound[(size_t)t], r1 = bound[(size_t)t + 1];
      if (r1 > r0) {
        memcpy(indptr + r0, ip + r0, sizeof(int32_t) * (size_t)(r1 - r0));
        const int64_t k0 = ip[r0], k1 = ip[r1];
        memcpy(indices + k0, ix + k0, sizeof(int32_t) * (size_t)(k1 - k0));
        if (da) memcpy(data + k0, da + k0, sizeof(double) * (size_t)(k1 - k0));
      }
    }
    indptr[rows] = (int32_t)nnz;
    return true;
  }
};

inline void spmv(const Csr& a, const double* x, double shift, const double* scale,
                 double* y) {
  // y[i] = (scale ? scale[i] : 1) * (shift + sum_k a[i,k] x[k]); every thread
  // walks the rows it first touched
  const int n_threads = (int)a.bound.size() - 1;
#pragma omp parallel num_threads(n_threads)
  {
    const int t = omp_get_thread_num();
    for (int64_t i = a.bound[(size_t)t]; i < a.bound[(size_t)t + 1]; ++i) {
      double acc = 0.;
      const int32_t b = a.indptr[i], e = a.indptr[i + 1];
      if (a.data) {
        for (int32_t k = b; k < e; ++k) acc += a.data[k] * x[a.indices[k]];
      } else {
        for (int32_t k = b; k < e; ++k) acc += x[a.indices[k]];
      }
      acc += shift;
      y[i] = scale ? scale[i] * acc : acc;
    }
  }
}

inline double dot(int64_t n, const double* a, const double* b) {
  double s = 0.;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

inline double sum(int64_t n, const double* a) {
  double s = 0.;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) s += a[i];
  return s;
}

struct Design {
  int64_t n = 0, p = 0;
  int intercept = 0, n_threads = 1;
  Csr x, xt;
  std::vector<double> offset;  // p
  ~Design() {
    x.release();
    xt.release();
  }
  // out[n] = rowscale .* (X~ v)
  void apply(const double* v, const double* rowscale, double* out) const {
    const double* v1 = v + intercept;
    double shift = intercept ? v[0] : 0.;
    shift -= dot(p, offset.data(), v1);
    spmv(x, v1, shift, rowscale, out);
  }
  // out[P] = X~^T w
  void apply_t(const double* w, double* out) const {
    const double sw = sum(n, w);
    if (intercept) out[0] = sw;
    double* g = out + intercept;
    spmv(xt, w, 0., nullptr, g);
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < p; ++j) g[j] -= sw * offset[(size_t)j];
  }
};

}  // namespace

extern "C" {

int oracle_omp_max_threads(void) { return omp_get_max_threads(); }

// Adopts (copies, with per-thread first touch) both orientations.  Returns an
// opaque handle or NULL.
void* oracle_omp_design_create(int64_t n, int64_t p, int intercept,
                               const int32_t* indptr, const int32_t* indices,
                               const double* data, const int32_t* t_indptr,
                               const int32_t* t_indices, const double* t_data,
                               const double* offset, int n_threads) {
  if (n_threads < 1) n_threads = omp_get_max_threads();
  omp_set_num_threads(n_threads);
  Design* d = new (std::nothrow) Design();
  if (!d) return nullptr;
  d->n = n;
  d->p = p;
  d->intercept = intercept;
  d->n_threads = n_threads;
  d->offset.assign(offset, offset + p);
  if (!d->x.adopt(n, indptr, indices, data, n_threads) ||
      !d->xt.adopt(p, t_indptr, t_indices, t_data, n_threads)) {
    delete d;
    return nullptr;
  }
  return d;
}

void oracle_omp_design_destroy(void* h) { delete static_cast<Design*>(h); }

// Plain products (for the baseline's own parity test and its GB/s figure).
int oracle_omp_dot(void* h, const double* v, double* out) {
  const Design* d = static_cast<const Design*>(h);
  omp_set_num_threads(d->n_threads);
  d->apply(v, nullptr, out);
  return 0;
}

int oracle_omp_tdot(void* h, const double* w, double* out) {
  const Design* d = static_cast<const Design*>(h);
  omp_set_num_threads(d->n_threads);
  d->apply_t(w, out);
  return 0;
}

// One draw of ConjugateGradientSampler.sample(precond_by='prior')
// (cg_sampler.py:20-94) with SciPy >= 1.14 `cg` semantics (M = I, stop when
// ||r|| < atol in preconditioned coordinates; x0.any() shortcut).
// Returns SciPy's info (0 converged, maxiter otherwise).
int oracle_omp_cg_sample(void* h, const double* omega, const double* phi,
                         const double* z, const double* x0, const double* sd,
                         int n_unshrunk, const double* eta1,
                         const double* eta2, int maxiter, double atol,
                         double* coef_out, int* n_iter_out) {
  const Design& D = *static_cast<const Design*>(h);
  omp_set_num_threads(D.n_threads);
  const int64_t n = D.n, P = D.p + D.intercept;
  std::vector<double> s(P), d(P), x(P), r(P), pv(P), q(P), sp(P), b(P), g(P);
  std::vector<double> t(n), w(n);
  for (int64_t j = 0; j < P; ++j) {                   // cg_sampler.py:128-138,104
    s[j] = j < n_unshrunk ? 2. * sd[j] : 1. / phi[j];
    const double a = s[j] * phi[j];
    d[j] = a * a;
    x[j] = x0[j] / s[j];
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) w[i] = std::sqrt(omega[i]) * eta1[i];
  D.apply_t(w.data(), g.data());                      // cg_sampler.py:66-68
  double bb = 0.;
  for (int64_t j = 0; j < P; ++j) {
    b[j] = s[j] * (z[j] + g[j] + phi[j] * eta2[j]);
    bb += b[j] * b[j];
  }
  (void)bb;  // rtol = atol/||b|| => stop at ||r|| < atol
  auto op = [&](const double* v, double* out) {       // cg_sampler.py:106-109
    for (int64_t j = 0; j < P; ++j) sp[j] = s[j] * v[j];
    D.apply(sp.data(), omega, t.data());
    D.apply_t(t.data(), g.data());
    for (int64_t j = 0; j < P; ++j) out[j] = d[j] * v[j] + s[j] * g[j];
  };
  bool any = false;
  for (int64_t j = 0; j < P; ++j) any = any || (x[j] != 0.);
  if (any) {
    op(x.data(), q.data());
    for (int64_t j = 0; j < P; ++j) r[j] = b[j] - q[j];
  } else {
    r = b;
  }
  double rho_prev = 1.;
  int k = 0, info = maxiter;
  for (; k < maxiter; ++k) {
    double rho = 0.;
    for (int64_t j = 0; j < P; ++j) rho += r[j] * r[j];
    if (std::sqrt(rho) < atol) {
      info = 0;
      break;
    }
    if (k > 0) {
      const double beta = rho / rho_prev;
      for (int64_t j = 0; j < P; ++j) pv[j] = r[j] + beta * pv[j];
    } else {
      pv = r;
    }
    op(pv.data(), q.data());
    double pq = 0.;
    for (int64_t j = 0; j < P; ++j) pq += pv[j] * q[j];
    const double alpha = rho / pq;
    for (int64_t j = 0; j < P; ++j) {
      x[j] += alpha * pv[j];
      r[j] -= alpha * q[j];
    }
    rho_prev = rho;
  }
  for (int64_t j = 0; j < P; ++j) coef_out[j] = s[j] * x[j];
  if (n_iter_out) *n_iter_out = k;
  return info;
}

}  // extern "C"
