// Scalar bodies of the solver families: wald, exgaussian (ppf by the reference's
// clipped Newton iteration find_ppf, optimization.py:10-32, run per lane with early exit), truncatednormal
// and skew_studentt (closed-form ppf).  Same Dist<D>::derive / eval<FN> template as m355_dists.h; every
// formula follows the reference line by line (file:line cited; paths relative to
// /root/reference/pytensor_distributions), its cancellations included (wald.cdf's e^{2 lam/mu} (1 + erf(-t))).
#pragma once
#include "m355_dists3.h"

namespace m355 {

#if defined(__HIPCC__)
M355_DEV double erfcinv_(double x) { return ::erfcinv(x); }
#else
M355_DEV double erfcinv_(double x) { return kNaN; }  // device libm only; not host-tested
#endif

// find_ppf (optimization.py:10-32): Newton steps clipped to max(|x|, 1), max_iter = 100, tol = 1e-8.  The
// reference iterates the whole batch until every element has converged; a converged element keeps x_prev,
// which makes it converge again in every later iteration, so a per-lane loop with early exit gives the same
// values.  A NaN iterate (NaN q, or a NaN cdf) stays NaN.
template <class CDF, class PDF>
M355_DEV_HEAVY double find_ppf(double q, double x0, double lower, double upper, CDF cdf, PDF pdf) {
  double x = x0;
  for (int it = 0; it < 100; ++it) {
    const double c = cdf(x);
    const double f0 = pdf(x);
    const double f = f0 > 1e-10 ? f0 : 1e-10;  // pt.maximum (a NaN pdf propagates: NaN > 1e-10 is false in
    double delta = (c - q) / ((f0 == f0) ? f : f0);  // the select, so put the NaN back)
    const double ax = __builtin_fabs(x);
    const double ms = ax > 1.0 ? ax : ((ax == ax) ? 1.0 : ax);
    delta = delta < -ms ? -ms : (delta > ms ? ms : delta);
    const double xn = x - delta;
    if (__builtin_fabs(xn - x) < 1e-8) break;  // converged: keeps x_prev
    x = xn;
    if (!(x == x)) break;
  }
  return ppf_bounds_cont(x, q, lower, upper);
}

// ----------------------------------------------------------------------------------------- wald
template <> struct Dist<M355_WALD> {
  static constexpr int NP = 2;
  struct K { double mu, lam, c, mode, e2; };
  static M355_HD K derive(const double* p) {
    const double ratio = 3.0 * p[0] / (2.0 * p[1]);  // wald.py:12-14
    return K{p[0], p[1], flog(p[1]) - flog(2.0 * kPi), p[0] * (::sqrt(1.0 + ratio * ratio) - ratio),
             ::exp(2.0 * p[1] / p[0])};
  }
  static M355_DEV double logpdf(double x, const K& k) {  // wald.py:77-82
    if ((x <= 0.0) || (x == kInf)) return -kInf;
    const double d = x - k.mu;
    return 0.5 * ((k.c - 3.0 * flog(x)) - k.lam * (d * d) / ((k.mu * k.mu) * x));
  }
  static M355_DEV double pdf(double x, const K& k) { return exp_(logpdf(x, k)); }  // wald.py:54-55
  static M355_DEV double cdf(double x, const K& k) {  // wald.py:40-47
    const double u = sqrt_(k.lam / (x + 1e-12));
    const double v = x / k.mu;
    const double z1 = 0.5 * (1.0 + erf_(u * (v - 1.0) / kSqrt2));
    const double z2 = k.e2 * 0.5 * (1.0 + erf_(-u * (v + 1.0) / kSqrt2));
    return cdf_bounds(z1 + z2, x, 0.0, kInf);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return pdf(x, k);
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_LOGCDF) return flog(cdf(x, k));          // wald.py:73-74
    if constexpr (FN == M355_LOGSF) return flog1p(-cdf(x, k));        // wald.py:85-86
    if constexpr (FN == M355_SF) return exp_(flog1p(-cdf(x, k)));     // wald.py:63-64
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                // wald.py:58-60, 50-51
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return find_ppf(q, k.mode, 0.0, kInf, [&](double v) { return cdf(v, k); }, [&](double v) { return pdf(v, k); });
    }
    return kNaN;
  }
};

// ----------------------------------------------------------------------------------- exgaussian
template <> struct Dist<M355_EXGAUSSIAN> {
  static constexpr int NP = 3;
  struct K { double mu, sigma, nu, lognu, half_sn2, shift, ex; };  // shift = mu + sigma^2/nu; ex = (nu >= 0.05 sigma)
  static M355_HD K derive(const double* p) {
    const double sn = p[1] / p[2];
    return K{p[0], p[1], p[2], flog(p[2]), 0.5 * (sn * sn), p[0] + (p[1] * p[1]) / p[2],
             (p[2] >= 0.05 * p[1]) ? 1.0 : 0.0};
  }
  static M355_DEV double logpdf(double x, const K& k) {  // exgaussian.py:101-111
    if (k.ex != 0.0) return ((-k.lognu + (k.mu - x) / k.nu) + k.half_sn2) + normal_logcdf_z(((x - k.shift) - 0.0) / k.sigma);
    const double z = (x - k.mu) / k.sigma;
    return (-0.5 * (z * z) - kLogSqrt2Pi) - flog(k.sigma);
  }
  static M355_DEV double logcdf(double x, const K& k) {  // exgaussian.py:76-98
    if (x == -kInf) return -kInf;
    if (x == kInf) return 0.0;
    const double a = normal_logcdf_z((x - k.mu) / k.sigma);
    if (k.ex == 0.0) return a;
    const double b = ((k.mu - x) / k.nu + k.half_sn2) + normal_logcdf_z(((x - k.shift) - 0.0) / k.sigma);
    return a + log1mexp_(b - a);  // helper.py:94-96 logdiffexp
  }
  static M355_DEV double cdf(double x, const K& k) { return exp_(logcdf(x, k)); }  // exgaussian.py:50-51
  static M355_DEV double pdf(double x, const K& k) { return exp_(logpdf(x, k)); }  // exgaussian.py:58-59
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return pdf(x, k);
    if constexpr (FN == M355_LOGCDF) return logcdf(x, k);
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);                    // exgaussian.py:66-67
    if constexpr (FN == M355_LOGSF) return flog1p(-exp_(logcdf(x, k)));     // exgaussian.py:114-115
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                      // exgaussian.py:62-63, 54-55
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return find_ppf(q, k.mu + k.nu, -kInf, kInf, [&](double v) { return cdf(v, k); },
                      [&](double v) { return pdf(v, k); });
    }
    return kNaN;
  }
};

// ------------------------------------------------------------------------------ truncatednormal
M355_DEV double tn_log_erfc(double x) { return flog(erfcx_(x)) - x * x; }  // truncatednormal.py:28-31
// log(Phi(beta) - Phi(alpha)), truncatednormal.py:34-61 (only the selected arm is evaluated)
M355_DEV double tn_log_Z(double alpha, double beta) {
  const double a = alpha / kSqrt2, b = beta / kSqrt2;
  if (alpha > 0.0) {
    const double la = tn_log_erfc(a), lb = tn_log_erfc(b);
    return kLogHalf + (la + log1mexp_(lb - la));
  }
  if (beta < 0.0) {
    const double la = tn_log_erfc(-b), lb = tn_log_erfc(-a);
    return kLogHalf + (la + log1mexp_(lb - la));
  }
  return kLogHalf + flog(erf_(b) - erf_(a));
}
template <> struct Dist<M355_TRUNCATEDNORMAL> {
  static constexpr int NP = 4;
  static constexpr bool kDeviceDerive = true;  // log Z needs erfcx: constants are derived on the device (LDS table)
  struct K { double mu, sigma, lo, hi, alpha, beta, logZ, c_norm; };
  static M355_DEV K derive(const double* p) {
    K k;
    k.mu = p[0];
    k.sigma = p[1];
    k.lo = p[2];
    k.hi = p[3];
    const double ar = (p[2] - p[0]) / p[1], br = (p[3] - p[0]) / p[1];  // truncatednormal.py:19-25
    k.alpha = (ar == -kInf) ? -100.0 : ar;
    k.beta = (br == kInf) ? 100.0 : br;
    k.logZ = tn_log_Z(k.alpha, k.beta);
    k.c_norm = -kLogSqrt2Pi - flog(p[1]);
    return k;
  }
  static M355_DEV double phi(double z) { return 0.5 * (1.0 + erf_(z / kSqrt2)); }  // truncatednormal.py:7-8
  static M355_DEV double cdf(double x, const K& k) {  // truncatednormal.py:186-206
    if (x <= k.lo) return 0.0;
    if (x >= k.hi) return 1.0;
    const double xi = (x - k.mu) / k.sigma;
    if (xi < k.alpha) return 0.0;
    if (xi > k.beta) return 1.0;
    return exp_(tn_log_Z(k.alpha, xi) - k.logZ);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF || FN == M355_PDF) {  // truncatednormal.py:172-183, 168-169
      const double z = (x - k.mu) / k.sigma;
      const double r = ((x < k.lo) || (x > k.hi)) ? -kInf : ((-0.5 * (z * z) + k.c_norm) - k.logZ);
      return FN == M355_PDF ? exp_(r) : r;
    }
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);  // truncatednormal.py:225-226
    if constexpr (FN == M355_LOGCDF) {                    // truncatednormal.py:209-222
      if (x <= k.lo) return -kInf;
      if (x >= k.hi) return 0.0;
      return tn_log_Z(k.alpha, (x - k.mu) / k.sigma) - k.logZ;
    }
    if constexpr (FN == M355_LOGSF) {  // truncatednormal.py:229-242
      if (x <= k.lo) return 0.0;
      if (x >= k.hi) return -kInf;
      return tn_log_Z((x - k.mu) / k.sigma, k.beta) - k.logZ;
    }
    if constexpr (FN == M355_PPF || FN == M355_ISF) {  // truncatednormal.py:245-266, 269-270
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      double r;
      if (k.alpha > 0.0) {
        const double sb = 0.5 * erfc_(k.beta / kSqrt2), sa = 0.5 * erfc_(k.alpha / kSqrt2);
        r = kSqrt2 * erfcinv_(2.0 * (q * sb + (1.0 - q) * sa));
      } else {
        const double pa = phi(k.alpha);
        r = kSqrt2 * erfinv_(2.0 * (q * (phi(k.beta) - pa) + pa) - 1.0);
      }
      return ppf_bounds_cont(k.mu + k.sigma * r, q, k.lo, k.hi);
    }
    return kNaN;
  }
};

// -------------------------------------------------------------------------------- skew_studentt
template <> struct Dist<M355_SKEW_STUDENTT> {
  static constexpr int NP = 4;
  struct K { double a, b, mu, sigma, ab, sqab, lbeta, c; };  // c = (a+b-1) log 2 + betaln + 0.5 log(a+b) + log sigma
  static M355_HD K derive(const double* p) {
    K k;
    k.a = p[0];
    k.b = p[1];
    k.mu = p[2];
    k.sigma = p[3];
    k.ab = p[0] + p[1];
    k.sqab = ::sqrt(k.ab);
    k.lbeta = lbeta_(p[0], p[1]);
    k.c = (((k.ab - 1.0) * kLog2 + k.lbeta) + 0.5 * flog(k.ab)) + flog(p[3]);
    return k;
  }
  static M355_DEV double logpdf(double x, const K& k) {  // skew_studentt.py:113-120
    if ((x == kInf) || (x == -kInf)) return kNaN;
    const double z = (x - k.mu) / k.sigma;
    const double s = sqrt_(k.ab + z * z);
    return ((k.a + 0.5) * flog(1.0 + z / s) + (k.b + 0.5) * flog(1.0 - z / s)) - k.c;
  }
  // cdf(x; a, b, mu, sigma), skew_studentt.py:123-129; sf / logsf call it with (b, a, -mu) (:141-147)
  static M355_DEV double cdf_raw(double x, double a, double b, double mu, const K& k) {
    if (x == kInf) return 1.0;
    if (x == -kInf) return 0.0;
    const double z = (x - mu) / k.sigma;
    const double y = 0.5 * (1.0 + z / sqrt_(k.ab + z * z));
    return betainc_(a, b, y, k.lbeta);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));  // skew_studentt.py:109-110
    if constexpr (FN == M355_CDF) return cdf_raw(x, k.a, k.b, k.mu, k);
    if constexpr (FN == M355_LOGCDF) return flog(cdf_raw(x, k.a, k.b, k.mu, k));  // skew_studentt.py:132-138
    if constexpr (FN == M355_SF) return cdf_raw(-x, k.b, k.a, -k.mu, k);
    if constexpr (FN == M355_LOGSF) return flog(cdf_raw(-x, k.b, k.a, -k.mu, k));
    if constexpr (FN == M355_PPF || FN == M355_ISF) {  // skew_studentt.py:150-157, 160-161
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      const double bv = betaincinv_(k.a, k.b, q, k.lbeta);
      const double z = ((2.0 * bv - 1.0) * k.sqab) / (2.0 * sqrt_(bv * (1.0 - bv)));
      return ppf_bounds_cont(k.mu + k.sigma * z, q, -kInf, kInf);
    }
    return kNaN;
  }
};

}  // namespace m355
