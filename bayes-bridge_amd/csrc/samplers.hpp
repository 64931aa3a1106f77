// Scalar samplers of the Gibbs steps that surround the CG draw, written once
// for host and device: Polya-Gamma (Omega update), exponentially tilted stable
// (local scales) and Gamma (global scale / observation variance).
//
// They are templates over a generator G providing `double uniform()` and
// `double normal()`.  Instantiated with bbx::Philox they run on the MI355X
// (distribution parity with the reference); instantiated on the host with
// NumPy's PCG64 bit generator they consume the stream in exactly the order of
// the reference's Cython code, which is what the exact-seed parity mode and
// the CPU tests against the reference rely on (hostrng.cpp).
//
// Algorithms follow (same random-number consumption order):
//   random/polya_gamma/polya_gamma.pyx:94-216  (Devroye alternating series for
//       the tilted Jacobi law; Polson, Scott & Windle 2013 eqs. 12-13;
//       Windle's thesis alg. 3 for the truncated inverse Gaussian)
//   random/tilted_stable/tilted_stable.pyx:136-331 (Hofert 2011 divide &
//       conquer; Devroye 2009 double rejection)
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define BBX_HD __host__ __device__
#else
#define BBX_HD
#endif

namespace bbx {

constexpr double kPi = 3.14159265358979323846;

// x^y for x > 0.  On the host this is libm pow, so that the exact-seed mode
// consumes and transforms NumPy's stream exactly like the reference's C code.
// On the device OCML's correctly rounded f64 pow costs several hundred
// dependent instructions and dominates the tilted-stable sampler (0.49 ms per
// Gibbs iteration at p = 5e4); exp(y log x) is ~3x cheaper and its relative
// error (~|y log x| * 2^-53 <= 1e-13 here) is irrelevant to a rejection
// sampler whose device stream only has to match in distribution.
BBX_HD inline double pos_pow(double x, double y) {
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(BBX_POS_POW_GENERIC)
  // The exponents of the tilted-stable sampler are functions of a = alpha / 2
  // alone -- a, 1 - a, 1 / (1 - a), (1 - a) / a, -1 / a -- i.e. constants of a
  // launch (uniform, scalar compares).  For the reference's default bridge
  // exponent alpha = 1/2 (a = 1/4: 1/4, 3/4, 4/3, +-3, +-4) and for alpha = 1
  // (a = 1/2: 1/2, +-1, +-2) every one of them is a root or a small integer
  // power: a few multiplications and square / cube roots, correctly rounded or
  // nearly so, instead of exp(y log x) -- 40 % of the lambda kernel's arithmetic
  // (profiles/r05_lscale_pow.txt).
  if (y == 0.25) return sqrt(sqrt(x));
  if (y == 0.75) {
    const double r = sqrt(x);
    return r * sqrt(r);
  }
  if (y == 0.5) return sqrt(x);
  if (y == 3.) return x * x * x;
  if (y == -3.) return 1. / (x * x * x);
  if (y == 4. / 3.) return x * cbrt(x);
  if (y == 4.) {
    const double q = x * x;
    return q * q;
  }
  if (y == -4.) {
    const double q = x * x;
    return 1. / (q * q);
  }
  if (y == 1.) return x;
  if (y == -1.) return 1. / x;
  if (y == 2.) return x * x;
  if (y == -2.) return 1. / (x * x);
#endif
  return exp(y * log(x));
#else
  return pow(x, y);
#endif
}

// log Phi(a), the log of the standard normal cdf (the reference calls the
// log_ndtr it vendors from SciPy, random/polya_gamma/scipy_ndtr.c:367, at
// polya_gamma.pyx:188-189).  Own formulation:
//   a >  6        Phi = 1 - Q, Q = erfc(a/sqrt2)/2 < 1e-9, log(1 - Q) = -Q to
//                 better than Q^2/2 < 1e-18;
//   -20 < a <= 6  log(erfc(-a/sqrt2)/2) directly: erfc does not underflow here;
//   a <= -20      Mills-ratio asymptotic expansion (Abramowitz & Stegun 26.2.12)
//                   Phi(a) = phi(a)/(-a) * sum_k (-1)^k (2k-1)!! / a^(2k),
//                 whose terms shrink by the factor (2k-1)/a^2 <= 1/400 * (2k-1);
//                 summed until a term no longer changes the sum.
BBX_HD inline double log_norm_cdf(double a) {
  constexpr double kInvSqrt2 = 0.70710678118654752440;
  constexpr double kHalfLog2Pi = 0.91893853320467274178;
  if (a > 6.) return -0.5 * erfc(a * kInvSqrt2);
  if (a > -20.) return log(0.5 * erfc(-a * kInvSqrt2));
  const double inv_a2 = 1. / (a * a);
  double series = 1., term = 1.;
  for (int k = 1; k <= 50; ++k) {
    term *= -(double)(2 * k - 1) * inv_a2;
    const double next = series + term;
    if (next == series) break;
    series = next;
  }
  return -0.5 * a * a - log(-a) - kHalfLog2Pi + log(series);
}

// ------------------------------------------------------------ Polya-Gamma

struct PolyaGamma {
  static constexpr double kCut = 2.0 / kPi;  // inverse-Gaussian | exponential
  static constexpr int kMaxTerms = 100;

  // n-th coefficient of the alternating series at x (polya_gamma.pyx:131-137).
  BBX_HD static inline double series_term(int n, double x) {
    const double h = n + 0.5;
    double lg = log(kPi * h);
    if (x <= kCut)
      lg += -1.5 * log(0.5 * x * kPi) - 2. * h * h / x;
    else
      lg += -0.5 * x * kPi * kPi * h * h;
    return exp(lg);
  }

  // Probability that the proposal comes from the exponential piece
  // (polya_gamma.pyx:115-128).
  BBX_HD static inline double right_mass(double z, double rate) {
    const double lm_exp = -log(rate) - rate * kCut + log(0.25 * kPi);
    const double sq = sqrt(kCut);
    const double lm_ig1 = -z + log_norm_cdf((kCut * z - 1.) / sq);
    const double lm_ig2 = z + log_norm_cdf(-(kCut * z + 1.) / sq);
    const double ratio = exp(lm_ig1 - lm_exp) + exp(lm_ig2 - lm_exp);
    return 1.0 / (1.0 + ratio);
  }

  template <class G>
  BBX_HD static inline double trunc_exp(G& g, double scale, double cut) {
    return cut - scale * log(1.0 - g.uniform());  // polya_gamma.pyx:164-165
  }

  // chi-square(1) restricted to [cut, inf) (polya_gamma.pyx:169-176).
  template <class G>
  BBX_HD static inline double trunc_chisq(G& g, double cut) {
    for (;;) {
      const double x = trunc_exp(g, 2., cut);
      const double ratio = sqrt(0.5 * kPi / x);
      if (g.uniform() <= ratio) return x;
    }
  }

  // Inverse Gaussian(mean, shape 1) (polya_gamma.pyx:200-207).
  template <class G>
  BBX_HD static inline double inv_gauss(G& g, double mean) {
    const double nrm = g.normal();
    const double v = nrm * nrm;
    double x = mean + 0.5 * mean *
                          (mean * v - sqrt(4.0 * mean * v + mean * mean * v * v));
    if (g.uniform() > mean / (mean + x)) x = mean * mean / x;
    return x;
  }

  // Inverse Gaussian(1/z, 1) restricted to (0, cut) (polya_gamma.pyx:179-198).
  template <class G>
  BBX_HD static inline double trunc_inv_gauss(G& g, double z, double cut) {
    const double mean = 1. / z;
    double x;
    if (mean > cut) {
      for (;;) {
        x = 1.0 / trunc_chisq(g, 0.5 * kPi);
        if (log(g.uniform()) < -0.5 * x * z * z) break;
      }
    } else {
      for (;;) {
        x = inv_gauss(g, mean);
        if (x < cut) break;
      }
    }
    return x;
  }

  // ---- the same sampler cut into pieces for the device's round structure
  // (chain.hip polya_gamma_block).  jacobi() below stays the sequential form
  // whose stream consumption is pinned to the reference's.

  // ONE proposal of trunc_inv_gauss: false = rejected (propose again).  The
  // nested loops above flatten to this: a rejection of the inner chi-square
  // proposal and a rejection of the outer test both lead to a fresh inner
  // proposal, i.e. to a fresh attempt.
  template <class G>
  BBX_HD static inline bool trunc_inv_gauss_attempt(G& g, double z, double cut,
                                                    double& x) {
    const double mean = 1. / z;
    if (mean > cut) {
      const double e = trunc_exp(g, 2., 0.5 * kPi);
      if (g.uniform() > sqrt(0.5 * kPi / e)) return false;
      x = 1.0 / e;
      return log(g.uniform()) < -0.5 * x * z * z;
    }
    x = inv_gauss(g, mean);
    return x < cut;
  }

  // The alternating-series test of a proposal x (the second half of one
  // iteration of jacobi()): true = accepted.
  template <class G>
  BBX_HD static inline bool series_accept(G& g, double x) {
    const double first = series_term(0, x);
    const double u = g.uniform() * first;
    double partial = first;
    int n_summed = 1;
    int sign = -1;
    for (;;) {
      partial += sign * series_term(n_summed, x);
      n_summed += 1;
      if (sign == -1) {
        if (u <= partial) return true;
      } else {
        if (u > partial) return false;
        if (n_summed >= kMaxTerms) return true;
      }
      sign = -sign;
    }
  }

  // ---- the device chain's forms of the two non-loop pieces (chain.hip
  // chain_pg_kernel): the same quantities without the detour through
  // logarithms, i.e. equal to right_mass / series_accept up to rounding --
  // which moves a draw only when a uniform falls within ~1e-15 of a threshold.
  // (The sequential jacobi() below, the host sampler pinned to the reference's
  // stream, keeps the reference's arithmetic.)

  // right_mass: mass_ig / mass_exp = (4 / pi) rate e^{rate kCut}
  //   [e^{-z} Phi((kCut z - 1)/sqrt kCut) + e^{z} Phi(-(kCut z + 1)/sqrt kCut)]
  // as products (two erfc, two exp instead of two erfc, three log, three exp);
  // from z = 20 on (|psi| >= 40) the factors leave the double range long
  // before their product does: the log form.
  BBX_HD static inline double right_mass_direct(double z, double rate) {
    if (z > 20.) return right_mass(z, rate);
    constexpr double kInvSqrt2 = 0.70710678118654752440;
    const double sq = sqrt(kCut);
    const double a = (kCut * z - 1.) / sq, b = -(kCut * z + 1.) / sq;
    const double c = rate * kCut;
    const double r1 = exp(c - z) * (0.5 * erfc(-a * kInvSqrt2));
    const double r2 = exp(c + z) * (0.5 * erfc(-b * kInvSqrt2));
    return 1.0 / (1.0 + (4. / kPi) * rate * (r1 + r2));
  }

  // series_accept with the terms a_n(x) = pi (n + 1/2) f(x) e^{-g(x)(n + 1/2)^2}
  // formed directly: f and g once per proposal, one exp per term.
  template <class G>
  BBX_HD static inline bool series_accept_direct(G& g, double x) {
    double f, gx;
    if (x <= kCut) {
      const double y = 0.5 * x * kPi;
      f = kPi / (y * sqrt(y));
      gx = 2. / x;
    } else {
      f = kPi;
      gx = 0.5 * x * kPi * kPi;
    }
    const double first = 0.5 * f * exp(-0.25 * gx);
    const double u = g.uniform() * first;
    double partial = first;
    int n_summed = 1;
    int sign = -1;
    for (;;) {
      const double h = n_summed + 0.5;
      partial += sign * (h * f * exp(-gx * h * h));
      n_summed += 1;
      if (sign == -1) {
        if (u <= partial) return true;
      } else {
        if (u > partial) return false;
        if (n_summed >= kMaxTerms) return true;
      }
      sign = -sign;
    }
  }

  // Tilted Jacobi J*(1, z) (polya_gamma.pyx:86-111,139-162).
  template <class G>
  BBX_HD static inline double jacobi(G& g, double z) {
    const double rate = 0.5 * z * z + 0.125 * kPi * kPi;
    for (;;) {
      const double p_right = right_mass(z, rate);
      double x;
      if (g.uniform() < p_right)
        x = trunc_exp(g, 1. / rate, kCut);
      else
        x = trunc_inv_gauss(g, z, kCut);
      const double first = series_term(0, x);
      const double u = g.uniform() * first;
      double partial = first;
      int n_summed = 1;
      int sign = -1;
      bool accepted = true;
      for (;;) {
        partial += sign * series_term(n_summed, x);
        n_summed += 1;
        if (sign == -1) {
          if (u <= partial) { accepted = true; break; }
        } else {
          if (u > partial) { accepted = false; break; }
          if (n_summed >= kMaxTerms) { accepted = true; break; }
        }
        sign = -sign;
      }
      if (accepted) return x;
    }
  }

  // PG(shape, tilt) for integer shape: sum of PG(1, tilt)
  // (polya_gamma.pyx:70-73,83-84).
  template <class G>
  BBX_HD static inline double draw(G& g, int shape, double tilt) {
    double acc = 0.;
    const double z = 0.5 * fabs(tilt);
    for (int j = 0; j < shape; ++j) acc += 0.25 * jacobi(g, z);
    return acc;
  }
};

// --------------------------------------------- exponentially tilted stable

struct TiltedStable {
  static constexpr double kMaxExpArg = 709.;
  static constexpr double kCostThreshold = 2.;  // tilted_stable.pyx:53

  BBX_HD static inline double safe_exp(double x) {  // tilted_stable.pyx:19-26
    if (x > kMaxExpArg) return INFINITY;
    if (x < -kMaxExpArg) return 0.;
    return exp(x);
  }

  BBX_HD static inline double sinc(double x) {  // tilted_stable.pyx:29-38
    if (fabs(x) < .01) {
      const double x2 = x * x;
      return 1. - x2 / 6. * (1 - x2 / 20.);
    }
    return sin(x) / x;
  }

  // Zolotarev's function A(x)^{1/(1-a)} (tilted_stable.pyx:324-331).
  BBX_HD static inline double zolotarev(double x, double a) {
    return pos_pow(pos_pow((1. - a) * sinc((1. - a) * x), (1. - a)) *
                   pos_pow(a * sinc(a * x), a) / sinc(x),
               1. / (1. - a));
  }

  // tilted_stable.pyx:313-322
  BBX_HD static inline double zolotarev_pdf_pow(double x, double a) {
    const double denom = pos_pow(sinc(a * x), a) * pos_pow(sinc((1. - a) * x), (1. - a));
    return sinc(x) / denom;
  }

  // Untilted positive stable via Kanter/Zolotarev (tilted_stable.pyx:156-163).
  template <class G>
  BBX_HD static inline double untilted(G& g, double a) {
    const double zf = zolotarev(kPi * g.uniform(), a);
    const double lg = log(g.uniform());
    return pos_pow(-zf / lg, (1. - a) / a);
  }

  // One proposal of the plain rejection sampler: S = c * (untilted stable),
  // accepted with probability exp(-tilt S) (tilted_stable.pyx:146-154).
  template <class G>
  BBX_HD static inline bool dc_trial(G& g, double a, double tilt, double c,
                                     double& s_out) {
    const double s = c * untilted(g, a);
    const double accept = safe_exp(-tilt * s);
    s_out = s;
    return g.uniform() < accept;
  }

  // Hofert's divide and conquer (tilted_stable.pyx:136-154).
  template <class G>
  BBX_HD static inline double divide_conquer(G& g, double a, double tilt) {
    long parts = (long)floor(pos_pow(tilt, a));
    if (parts < 1) parts = 1;
    const double c = pos_pow(1. / parts, 1. / a);
    double x = 0.;
    for (long i = 0; i < parts; ++i) {
      for (;;) {
        double s;
        if (dc_trial(g, a, tilt, c, s)) { x += s; break; }
      }
    }
    return x;
  }

  // tilted_stable.pyx:216-241
  template <class G>
  BBX_HD static inline double aux2(G& g, double xi, double psi, double gamma) {
    const double w1 = sqrt(.5 * kPi / gamma) * xi;
    const double w2 = 2. * sqrt(kPi) * psi;
    const double w3 = xi * kPi;
    const double v = g.uniform();
    double u;
    if (gamma >= 1) {
      if (v < w1 / (w1 + w2)) {
        u = fabs(g.normal()) / sqrt(gamma);
      } else {
        const double w = g.uniform();
        u = kPi * (1. - w * w);
      }
    } else {
      const double w = g.uniform();
      if (v < w3 / (w2 + w3))
        u = kPi * w;
      else
        u = kPi * (1. - w * w);
    }
    return u;
  }

  // tilted_stable.pyx:243-259
  BBX_HD static inline double aux2_accept(double u, double xi, double psi,
                                          double zeta, double z,
                                          double tilt_pow, double gamma) {
    double inv = kPi * safe_exp(-tilt_pow * (1. - 1. / (zeta * zeta))) /
                 ((1. + sqrt(.5 * kPi)) * sqrt(gamma) / zeta + z);
    double d = 0.;
    if (u >= 0. && gamma >= 1) d += xi * safe_exp(-gamma * u * u / 2.);
    if (u > 0. && u < kPi) d += psi / sqrt(kPi - u);
    if (u >= 0. && u <= kPi && gamma < 1.) d += xi;
    inv *= d;
    return 1 / inv;
  }

  // One outer iteration of Devroye's double rejection
  // (tilted_stable.pyx:165-311): auxiliary variable U (inner rejection loop),
  // reference variable X | U, acceptance test.  Returns the candidate through
  // x_out (to be transformed by pos_pow(x, -(1-a)/a) when accepted).
  template <class G>
  BBX_HD static inline bool dr_trial(G& g, double a, double tilt_pow,
                                     double& x_out) {
    const double odds = (1. - a) / a;
    // --- auxiliary variable U (tilted_stable.pyx:181-214)
    const double gamma = tilt_pow * a * (1. - a);
    const double xi = (1. + sqrt(2. * gamma) * (2. + sqrt(.5 * kPi))) / kPi;
    const double psi = sqrt(gamma / kPi) * (2. + sqrt(.5 * kPi)) *
                       safe_exp(-gamma * kPi * kPi / 8.);
    double u, v, z;
    for (;;) {
      u = aux2(g, xi, psi, gamma);
      if (u > kPi) continue;
      const double zeta = sqrt(zolotarev_pdf_pow(u, a));
      z = 1. / (1. - pos_pow(1. + a * zeta / sqrt(gamma), -1. / a));
      const double ap = aux2_accept(u, xi, psi, zeta, z, tilt_pow, gamma);
      if (ap > 0.) {
        v = g.uniform() / ap;
        if (u < kPi && v <= 1.) break;
      }
    }
    // --- reference variable X | U (tilted_stable.pyx:261-296)
    const double aa = zolotarev(u, a);
    const double left = pos_pow(odds / aa, a) * tilt_pow;
    const double right = left + sqrt(left * a / aa);
    const double expo_scale = z / aa;
    const double m_left = (right - left) * sqrt(.5 * kPi);
    const double m_mid = (right - left);
    const double m_right = expo_scale;
    const double m_tot = m_left + m_mid + m_right;
    const double pick = g.uniform();
    double nrm = 0., e = 0., x;
    if (pick < m_left / m_tot) {
      nrm = g.normal();
      x = left - (right - left) * fabs(nrm);
    } else if (pick < (m_left + m_mid) / m_tot) {
      x = left + (right - left) * g.uniform();
    } else {
      e = -log(g.uniform());
      x = right + e * m_right;
    }
    // --- acceptance (tilted_stable.pyx:298-311)
    double log_accept;
    if (x < 0) {
      log_accept = -INFINITY;
    } else {
      log_accept = -(aa * (x - left) +
                     safe_exp(log(tilt_pow) / a - odds * log(left)) *
                         (pos_pow(left / x, odds) - 1.));
      if (x < left)
        log_accept += nrm * nrm / 2.;
      else if (x > right)
        log_accept += e;
    }
    x_out = x;
    return log_accept > log(v);
  }

  // The same proposal with the INNER rejection loop unrolled into the outer
  // one: a candidate U that fails the auxiliary test ends the trial (the
  // caller starts a fresh one) instead of being redrawn in place.  Attempts
  // are independent and identically distributed, so "first attempt whose U
  // and X both pass" has exactly the law of dr_trial's "first accepted X among
  // inner-accepted U's"; only the bookkeeping differs.  On a wavefront the
  // in-place loop runs as long as its unluckiest lane (~12 passes for 64 lanes
  // at the ~30 % inner acceptance of tilt^a in [2, 8]); flattened, every lane
  // does one short pass and failed items retry with the block's spare lanes.
  template <class G>
  BBX_HD static inline bool dr_trial_flat(G& g, double a, double tilt_pow,
                                          double& x_out) {
    const double odds = (1. - a) / a;
    const double gamma = tilt_pow * a * (1. - a);
    const double xi = (1. + sqrt(2. * gamma) * (2. + sqrt(.5 * kPi))) / kPi;
    const double psi = sqrt(gamma / kPi) * (2. + sqrt(.5 * kPi)) *
                       safe_exp(-gamma * kPi * kPi / 8.);
    const double u = aux2(g, xi, psi, gamma);
    if (!(u < kPi)) return false;
    const double zeta = sqrt(zolotarev_pdf_pow(u, a));
    const double z = 1. / (1. - pos_pow(1. + a * zeta / sqrt(gamma), -1. / a));
    const double ap = aux2_accept(u, xi, psi, zeta, z, tilt_pow, gamma);
    if (!(ap > 0.)) return false;
    const double v = g.uniform() / ap;
    if (!(v <= 1.)) return false;
    // --- reference variable X | U and the acceptance test, as in dr_trial
    const double aa = zolotarev(u, a);
    const double left = pos_pow(odds / aa, a) * tilt_pow;
    const double right = left + sqrt(left * a / aa);
    const double expo_scale = z / aa;
    const double m_left = (right - left) * sqrt(.5 * kPi);
    const double m_mid = (right - left);
    const double m_right = expo_scale;
    const double m_tot = m_left + m_mid + m_right;
    const double pick = g.uniform();
    double nrm = 0., e = 0., x;
    if (pick < m_left / m_tot) {
      nrm = g.normal();
      x = left - (right - left) * fabs(nrm);
    } else if (pick < (m_left + m_mid) / m_tot) {
      x = left + (right - left) * g.uniform();
    } else {
      e = -log(g.uniform());
      x = right + e * m_right;
    }
    if (x < 0) return false;
    double log_accept =
        -(aa * (x - left) + safe_exp(log(tilt_pow) / a - odds * log(left)) *
                                (pos_pow(left / x, odds) - 1.));
    if (x < left)
      log_accept += nrm * nrm / 2.;
    else if (x > right)
      log_accept += e;
    x_out = x;
    return log_accept > log(v);
  }

  // Devroye's double rejection (tilted_stable.pyx:165-311).
  template <class G>
  BBX_HD static inline double double_rejection(G& g, double a, double tilt) {
    const double tilt_pow = pos_pow(tilt, a);
    const double odds = (1. - a) / a;
    for (;;) {
      double x;
      if (dr_trial(g, a, tilt_pow, x)) return pos_pow(x, -odds);
    }
  }

  // Method choice of tilted_stable.pyx:99-104.
  template <class G>
  BBX_HD static inline double draw(G& g, double a, double tilt) {
    if (pos_pow(tilt, a) < kCostThreshold) return divide_conquer(g, a, tilt);
    return double_rejection(g, a, tilt);
  }
};

// ------------------------------------------------------------------ Gamma

// Gamma(shape, 1), Marsaglia & Tsang (2000).  Device-only use (the reference
// draws these two scalars per iteration with np.random.gamma,
// bayesbridge.py:403,438); distribution parity.
template <class G>
BBX_HD inline double gamma_draw(G& g, double shape) {
  double boost = 1.;
  if (shape < 1.) {
    boost = pow(g.uniform(), 1. / shape);
    shape += 1.;
  }
  const double d = shape - 1. / 3.;
  const double c = 1. / sqrt(9. * d);
  for (;;) {
    double x, v;
    do {
      x = g.normal();
      v = 1. + c * x;
    } while (v <= 0.);
    v = v * v * v;
    const double u = g.uniform();
    if (u < 1. - 0.0331 * (x * x) * (x * x)) return boost * d * v;
    if (log(u) < 0.5 * x * x + d * (1. - v + log(v))) return boost * d * v;
  }
}

}  // namespace bbx
