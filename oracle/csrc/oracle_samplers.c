/*
 * oracle_samplers.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C restatement of the two scalar rejection samplers the reference
 * keeps in Cython, so that the CPU oracle can run whole Gibbs chains at the
 * reference's speed and on the reference's random streams:
 *
 *   oracle_polya_gamma      random/polya_gamma/polya_gamma.pyx:40-216
 *   oracle_tilted_stable    random/tilted_stable/tilted_stable.pyx:65-331
 *
 * Both take the address of a NumPy bitgen_t (PCG64(seed).ctypes.bit_generator)
 * and draw uniforms with next_double and normals with libnpyrandom's
 * random_standard_normal, exactly as random/uniform/uniform.pyx and
 * random/normal/normal.pyx do (setup.py:12-13,34-46 link the same library).
 * log Phi uses libm erfc above -20 and the published asymptotic series below
 * (the reference vendors Cephes' log_ndtr, random/polya_gamma/scipy_ndtr.c:367).
 *
 * Built by oracle/Makefile into oracle/_build/liboracle.so; pinned in
 * tests/test_oracle_vs_reference.py against the reference's own samplers
 * (same seeds => same draws and same final generator state).
 */
#include <math.h>
#include <stdint.h>

typedef struct bitgen {
  void *state;
  uint64_t (*next_uint64)(void *st);
  uint32_t (*next_uint32)(void *st);
  double (*next_double)(void *st);
  uint64_t (*next_raw)(void *st);
} bitgen_t;

extern double random_standard_normal(bitgen_t *bitgen_state);

#define PI 3.14159265358979323846
#define PG_CUT (2.0 / PI)
#define PG_MAX_TERMS 100

static double unif(bitgen_t *bg) { return bg->next_double(bg->state); }

static double log_phi(double a) {
  double lhs, last = 0.0, rhs = 1.0, num = 1.0, den = 1.0, inv;
  long sign = 1, i = 0;
  if (a > 6.0) return -0.5 * erfc(a / sqrt(2.0));
  if (a > -20.0) return log(0.5 * erfc(-a / sqrt(2.0)));
  lhs = -0.5 * a * a - log(-a) - 0.5 * log(2.0 * PI);
  inv = 1.0 / (a * a);
  while (fabs(last - rhs) > 2.220446049250313e-16 && i < 60) {
    i += 1;
    last = rhs;
    sign = -sign;
    den *= inv;
    num *= (double)(2 * i - 1);
    rhs += (double)sign * num * den;
  }
  return lhs + log(rhs);
}

/* ---------------------------------------------------------- Polya-Gamma */

static double pg_coef(int n, double x) { /* polya_gamma.pyx:131-137 */
  double k = n + 0.5, lr = log(PI * k);
  if (x <= PG_CUT)
    lr += -1.5 * log(0.5 * x * PI) - 2.0 * k * k / x;
  else
    lr += -0.5 * x * PI * PI * k * k;
  return exp(lr);
}

static double pg_prob_right(double tilt, double rate) { /* :115-128 */
  double le = -log(rate) - rate * PG_CUT + log(0.25 * PI);
  double l1 = -tilt + log_phi((PG_CUT * tilt - 1.0) / sqrt(PG_CUT));
  double l2 = tilt + log_phi(-(PG_CUT * tilt + 1.0) / sqrt(PG_CUT));
  return 1.0 / (1.0 + exp(l1 - le) + exp(l2 - le));
}

static double pg_left_trunc_exp(bitgen_t *bg, double scale, double trunc) {
  return trunc - scale * log(1.0 - unif(bg)); /* :164-165 */
}

static double pg_left_trunc_chisq(bitgen_t *bg, double trunc) { /* :169-176 */
  for (;;) {
    double x = pg_left_trunc_exp(bg, 2.0, trunc);
    if (unif(bg) <= sqrt(0.5 * PI / x)) return x;
  }
}

static double pg_invgauss(bitgen_t *bg, double mean) { /* :200-207 */
  double z = random_standard_normal(bg);
  double v = z * z;
  double x = mean + 0.5 * mean * (mean * v - sqrt(4.0 * mean * v + mean * mean * v * v));
  if (unif(bg) > mean / (mean + x)) x = mean * mean / x;
  return x;
}

static double pg_right_trunc_invgauss(bitgen_t *bg, double rate, double trunc) {
  double mean = 1.0 / rate, x; /* :179-198 */
  if (mean > trunc) {
    do {
      x = 1.0 / pg_left_trunc_chisq(bg, 0.5 * PI);
    } while (!(log(unif(bg)) < -0.5 * x * rate * rate));
  } else {
    do {
      x = pg_invgauss(bg, mean);
    } while (!(x < trunc));
  }
  return x;
}

static double pg_tilted_jacobi(bitgen_t *bg, double tilt) { /* :86-162 */
  for (;;) {
    double rate = 0.5 * tilt * tilt + 0.125 * PI * PI;
    double x, a0, u, s;
    int n = 1, sign = -1, verdict = -1;
    if (unif(bg) < pg_prob_right(tilt, rate))
      x = pg_left_trunc_exp(bg, 1.0 / rate, PG_CUT);
    else
      x = pg_right_trunc_invgauss(bg, tilt, PG_CUT);
    a0 = pg_coef(0, x);
    u = unif(bg) * a0;
    s = a0;
    while (verdict < 0) {
      s += sign * pg_coef(n, x);
      n += 1;
      if (sign == -1) {
        if (u <= s) verdict = 1;
      } else {
        if (u > s) verdict = 0;
        else if (n >= PG_MAX_TERMS) verdict = 1;
      }
      sign = -sign;
    }
    if (verdict) return x;
  }
}

int oracle_polya_gamma(void *bitgen, int64_t n, const int32_t *shape,
                       const double *tilt, double *out) {
  bitgen_t *bg = (bitgen_t *)bitgen;
  int64_t i;
  int j;
  for (i = 0; i < n; ++i) { /* polya_gamma.pyx:70-73 */
    double acc = 0.0;
    for (j = 0; j < shape[i]; ++j)
      acc += 0.25 * pg_tilted_jacobi(bg, 0.5 * fabs(tilt[i]));
    out[i] = acc;
  }
  return 0;
}

/* -------------------------------------------------------- tilted stable */

static double ts_exp(double x) { /* tilted_stable.pyx:19-26 */
  if (x > 709.0) return INFINITY;
  if (x < -709.0) return 0.0;
  return exp(x);
}

static double ts_sinc(double x) { /* :29-38 */
  if (fabs(x) < 0.01) {
    double q = x * x;
    return 1.0 - q / 6.0 * (1 - q / 20.0);
  }
  return sin(x) / x;
}

static double ts_zolotarev(double x, double a) { /* :324-331 */
  return pow(pow((1.0 - a) * ts_sinc((1.0 - a) * x), 1.0 - a) *
                 pow(a * ts_sinc(a * x), a) / ts_sinc(x),
             1.0 / (1.0 - a));
}

static double ts_zolotarev_pdf(double x, double a) { /* :313-322 */
  return ts_sinc(x) /
         (pow(ts_sinc(a * x), a) * pow(ts_sinc((1.0 - a) * x), 1.0 - a));
}

static double ts_divide_conquer(bitgen_t *bg, double a, double tilt) {
  long m = (long)floor(pow(tilt, a)), i; /* :136-163 */
  double c, total = 0.0;
  if (m < 1) m = 1;
  c = pow(1.0 / m, 1.0 / a);
  for (i = 0; i < m; ++i) {
    for (;;) {
      double zf = ts_zolotarev(PI * unif(bg), a);
      double lg = log(unif(bg));
      double s = c * pow(-zf / lg, (1.0 - a) / a);
      if (unif(bg) < ts_exp(-tilt * s)) {
        total += s;
        break;
      }
    }
  }
  return total;
}

static double ts_aux2(bitgen_t *bg, double xi, double psi, double gam) {
  double w1 = sqrt(0.5 * PI / gam) * xi; /* :216-241 */
  double w2 = 2.0 * sqrt(PI) * psi;
  double w3 = xi * PI;
  double v = unif(bg), w;
  if (gam >= 1) {
    if (v < w1 / (w1 + w2)) return fabs(random_standard_normal(bg)) / sqrt(gam);
    w = unif(bg);
    return PI * (1.0 - w * w);
  }
  w = unif(bg);
  if (v < w3 / (w2 + w3)) return PI * w;
  return PI * (1.0 - w * w);
}

static double ts_double_rejection(bitgen_t *bg, double a, double tilt) {
  double tp = pow(tilt, a), odds = (1.0 - a) / a; /* :165-311 */
  for (;;) {
    double gam = tp * a * (1.0 - a);
    double xi = (1.0 + sqrt(2.0 * gam) * (2.0 + sqrt(0.5 * PI))) / PI;
    double psi = sqrt(gam / PI) * (2.0 + sqrt(0.5 * PI)) *
                 ts_exp(-gam * PI * PI / 8.0);
    double u, v = 0.0, z = 0.0, aa, lo, hi, ml, mm, mr, mt, pick, x;
    double nrm = 0.0, e = 0.0, la;
    for (;;) {
      double zeta, inv, d = 0.0, ap;
      u = ts_aux2(bg, xi, psi, gam);
      if (u > PI) continue;
      zeta = sqrt(ts_zolotarev_pdf(u, a));
      z = 1.0 / (1.0 - pow(1.0 + a * zeta / sqrt(gam), -1.0 / a));
      inv = PI * ts_exp(-tp * (1.0 - 1.0 / (zeta * zeta))) /
            ((1.0 + sqrt(0.5 * PI)) * sqrt(gam) / zeta + z);
      if (u >= 0.0 && gam >= 1) d += xi * ts_exp(-gam * u * u / 2.0);
      if (u > 0.0 && u < PI) d += psi / sqrt(PI - u);
      if (u >= 0.0 && u <= PI && gam < 1.0) d += xi;
      ap = 1 / (inv * d);
      if (ap > 0.0) {
        v = unif(bg) / ap;
        if (u < PI && v <= 1.0) break;
      }
    }
    aa = ts_zolotarev(u, a);
    lo = pow(odds / aa, a) * tp;
    hi = lo + sqrt(lo * a / aa);
    ml = (hi - lo) * sqrt(0.5 * PI);
    mm = hi - lo;
    mr = z / aa;
    mt = ml + mm + mr;
    pick = unif(bg);
    if (pick < ml / mt) {
      nrm = random_standard_normal(bg);
      x = lo - (hi - lo) * fabs(nrm);
    } else if (pick < (ml + mm) / mt) {
      x = lo + (hi - lo) * unif(bg);
    } else {
      e = -log(unif(bg));
      x = hi + e * mr;
    }
    if (x < 0) {
      la = -INFINITY;
    } else {
      la = -(aa * (x - lo) + ts_exp(log(tp) / a - odds * log(lo)) *
                                 (pow(lo / x, odds) - 1.0));
      if (x < lo) la += nrm * nrm / 2.0;
      else if (x > hi) la += e;
    }
    if (la > log(v)) return pow(x, -odds);
  }
}

int oracle_tilted_stable(void *bitgen, int64_t n, double char_exp,
                         const double *tilt, double *out) {
  bitgen_t *bg = (bitgen_t *)bitgen;
  int64_t i;
  for (i = 0; i < n; ++i) { /* method choice: tilted_stable.pyx:99-104 */
    if (pow(tilt[i], char_exp) < 2.0)
      out[i] = ts_divide_conquer(bg, char_exp, tilt[i]);
    else
      out[i] = ts_double_rejection(bg, char_exp, tilt[i]);
  }
  return 0;
}

/* ------------------------------------------------- CSR products (baseline)
 * Restatement of the two SciPy kernels the reference spends its time in
 * (scipy/sparse/sparsetools/csr.h csr_matvec; csc.h csc_matvec applied to the
 * CSC view of X.T, i.e. a row scatter), used to cross-check the NumPy path. */
int oracle_csr_matvec(int64_t n_row, const int32_t *indptr,
                      const int32_t *indices, const double *data,
                      const double *x, double *y) {
  int64_t i;
  int32_t k;
  for (i = 0; i < n_row; ++i) {
    double sum = y[i];
    for (k = indptr[i]; k < indptr[i + 1]; ++k) sum += data[k] * x[indices[k]];
    y[i] = sum;
  }
  return 0;
}

int oracle_csr_rmatvec(int64_t n_row, const int32_t *indptr,
                       const int32_t *indices, const double *data,
                       const double *w, double *y) {
  int64_t i;
  int32_t k;
  for (i = 0; i < n_row; ++i)
    for (k = indptr[i]; k < indptr[i + 1]; ++k) y[indices[k]] += data[k] * w[i];
  return 0;
}
