// Scalar bodies of the solver and series families: wald, exgaussian, skewnormal (Owen's T),
// vonmises (Bessel-ratio series), rice (scaled Bessel functions, ncx2 / Marcum-Q sum) -- ppf by the reference's
// clipped Newton iteration find_ppf, optimization.py:10-32, run per lane with early exit -- polyagamma (series
// density, trapezoid cdf, its own Newton ppf), and truncatednormal, skew_studentt (closed-form ppf).  Same Dist<D>::derive / eval<FN> template as m355_dists.h; every
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
#ifndef M355_FIND_PPF_ATTR
#define M355_FIND_PPF_ATTR M355_DEV_HEAVY  // tools/findppf_probe.hip builds the inlined variant for comparison
#endif
template <class CDF, class PDF>
M355_FIND_PPF_ATTR double find_ppf(double q, double x0, double lower, double upper, CDF cdf, PDF pdf) {
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

// ------------------------------------------------------------------------------------ Owen's T
// T(h, a) = (1/2pi) int_0^a exp(-h^2 (1 + x^2)/2) / (1 + x^2) dx (pt.owens_t = scipy.special.owens_t).
// |a| <= 1: 32-point Gauss-Legendre on [0, a] (the integrand is entire with width >= 1/|h|; measured
// |error| <= 1e-16 against SciPy for |h| <= 12); |a| > 1: Owen's reflection
// T(h, a) = (Q(h) + Q(ah))/2 - Q(h) Q(ah) - T(ah, 1/a), Q = erfc(./sqrt2)/2 (h >= 0), which has no cancellation.
M355_DEV double owens_t_gl(double h, double a) {  // h >= 0, 0 <= a <= 1
  constexpr double kNode[32] = {1.36806907525921506e-03, 7.19424422736580915e-03, 1.76188722062468051e-02, 3.25469620311301666e-02, 5.18394221169738989e-02, 7.53161931337150148e-02, 1.02758102016028807e-01, 1.33908940629855144e-01, 1.68477866534892384e-01, 2.06142121379618848e-01, 2.46550045533885320e-01, 2.89324361934682361e-01, 3.34065698858936166e-01, 3.80356318873931454e-01, 4.27764019208601742e-01, 4.75846167156130817e-01, 5.24153832843869183e-01, 5.72235980791398258e-01, 6.19643681126068491e-01, 6.65934301141063889e-01, 7.10675638065317639e-01, 7.53449954466114624e-01, 7.93857878620381152e-01, 8.31522133465107616e-01, 8.66091059370144856e-01, 8.97241897983971137e-01, 9.24683806866285041e-01, 9.48160577883026101e-01, 9.67453037968869833e-01, 9.82381127793753195e-01, 9.92805755772634191e-01, 9.98631930924740785e-01};
  constexpr double kWeight[32] = {3.50930500473464919e-03, 8.13719736545298264e-03, 1.26960326546312133e-02, 1.71369314565108129e-02, 2.14179490111132131e-02, 2.54990296311881220e-02, 2.93420467392678519e-02, 3.29111113881808762e-02, 3.61728970544242245e-02, 3.90969478935351555e-02, 4.16559621134734229e-02, 4.38260465022019541e-02, 4.55869393478819315e-02, 4.69221995404022832e-02, 4.78193600396374166e-02, 4.82700442573639060e-02, 4.82700442573639060e-02, 4.78193600396374166e-02, 4.69221995404022832e-02, 4.55869393478819315e-02, 4.38260465022019541e-02, 4.16559621134734229e-02, 3.90969478935351555e-02, 3.61728970544242245e-02, 3.29111113881808762e-02, 2.93420467392678519e-02, 2.54990296311881220e-02, 2.14179490111132131e-02, 1.71369314565108129e-02, 1.26960326546312133e-02, 8.13719736545298264e-03, 3.50930500473464919e-03};
  const double hh = -0.5 * h * h;
  double s = 0.0;
#pragma unroll 4
  for (int i = 0; i < 32; ++i) {
    const double x = a * kNode[i];
    const double d = 1.0 + x * x;
    s += kWeight[i] * exp_(hh * d) / d;
  }
  return s * a * (0.5 / kPi);
}
M355_DEV_HEAVY double owens_t_(double h, double a) {
  if (!(h == h) || !(a == a)) return kNaN;
  const double sgn = a < 0.0 ? -1.0 : 1.0;
  a = __builtin_fabs(a);
  h = __builtin_fabs(h);
  if (a == 0.0 || h == kInf) return sgn * 0.0;
  if (a <= 1.0) return sgn * owens_t_gl(h, a);
  const double q1 = 0.5 * erfc_(h * kInvSqrt2);
  if (a == kInf) return sgn * 0.5 * q1;
  const double ah = a * h;
  const double q2 = 0.5 * erfc_(ah * kInvSqrt2);
  return sgn * ((0.5 * (q1 + q2) - q1 * q2) - owens_t_gl(ah, 1.0 / a));
}

// ----------------------------------------------------------------------------------------- wald
template <> struct Dist<M355_WALD> {
  static constexpr int kBinned = 2;
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
  static constexpr int kBinned = 2;
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
  static constexpr int kMinWaves = 3;  // binned kernel: 168-VGPR budget around the inlined betaincinv_ (m355_elemwise.h)
  static constexpr int kBinned = 1;
  static constexpr int NP = 4;
  struct K { double a, b, mu, sigma, ab, sqab, lbeta, c, ihalf; };  // c = (a+b-1) log 2 + betaln + 0.5 log(a+b) + log sigma; ihalf = I_{1/2}(a, b)
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
    k.ihalf = betainc_half(p[0], p[1], k.lbeta);
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
      const double bv = betaincinv_(k.a, k.b, q, k.lbeta, k.ihalf);
      const double z = ((2.0 * bv - 1.0) * k.sqab) / (2.0 * sqrt_(bv * (1.0 - bv)));
      return ppf_bounds_cont(k.mu + k.sigma * z, q, -kInf, kInf);
    }
    return kNaN;
  }
};

// ----------------------------------------------------------------------------------- skewnormal
template <> struct Dist<M355_SKEWNORMAL> {
  static constexpr int kBinned = 2;
  static constexpr int NP = 3;
  struct K { double mu, sigma, alpha, sqrt_tau, tau, c, mean; };  // tau = 1/sigma^2; c = log(tau/pi/2)
  static M355_HD K derive(const double* p) {
    const double tau = 1.0 / (p[1] * p[1]);
    return K{p[0], p[1], p[2], ::sqrt(tau), tau, flog(tau / kPi / 2.0),
             p[0] + p[1] * ::sqrt(2.0 / kPi) * p[2] / ::sqrt(1.0 + p[2] * p[2])};  // skewnormal.py:8-10
  }
  static M355_DEV double logpdf(double x, const K& k) {  // skewnormal.py:95-100
    const double d = x - k.mu;
    return flog(1.0 + erf_((d * k.sqrt_tau * k.alpha) / kSqrt2)) + (-k.tau * (d * d) + k.c) / 2.0;
  }
  static M355_DEV double pdf(double x, const K& k) { return exp_(logpdf(x, k)); }  // skewnormal.py:73-74
  static M355_DEV double cdf(double x, const K& k) {  // skewnormal.py:61-62
    return 0.5 * (1.0 + erf_((x - k.mu) / (k.sigma * kSqrt2))) - 2.0 * owens_t_((x - k.mu) / k.sigma, k.alpha);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return pdf(x, k);
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_LOGCDF) return flog(cdf(x, k));             // skewnormal.py:65-66
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);                 // skewnormal.py:83-84
    if constexpr (FN == M355_LOGSF) return log1mexp_(flog(cdf(x, k)));   // skewnormal.py:103-104
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                   // skewnormal.py:77-80, 69-70
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return find_ppf(q, k.mean, -kInf, kInf, [&](double v) { return cdf(v, k); }, [&](double v) { return pdf(v, k); });
    }
    return kNaN;
  }
};

// ------------------------------------------------------------------------------------- vonmises
// Everything Bessel is a function of kappa only and lives in derive(): the ratios I_n(kappa)/I_0(kappa),
// n <= 50, by the backward recurrence f_{n-1} = (2n/kappa) f_n + f_{n+1} (the stable direction, started at
// n = 130 and rescaled against overflow), normalised with e^kappa = I_0 + 2 sum_n I_n, which also gives
// ive(0, kappa); for kappa >= 50 (normal approximation of the cdf, helper.py:365-368) ive(0, kappa) comes
// from its asymptotic series.  Per element: helper.py:337-372 with the series coefficients read from K.
template <> struct Dist<M355_VONMISES> {
  static constexpr int kBinned = 2;
  static constexpr int NP = 2;
  static constexpr int kTerms = 50;
  struct K {
    double mu, kappa, log_i0e, b, p;  // b = sqrt(2/pi)/ive(0, kappa); p = number of series terms
    double coef[kTerms];             // I_n/I_0 / n, n = 1..50
  };
  static M355_HD K derive(const double* pp) {
    K k;
    k.mu = pp[0];
    k.kappa = pp[1];
    const double kappa = pp[1];
    double i0e;
    for (int n = 0; n < kTerms; ++n) k.coef[n] = 0.0;
    if (kappa == 0.0) {
      i0e = 1.0;
    } else if (kappa < 50.0) {
      const int N = 130;
      double fp = 0.0, f = 1e-280, sum = 0.0;  // f_{n+1}, f_n
      for (int n = N; n >= 1; --n) {
        sum += f;
        if (n <= kTerms) k.coef[n - 1] = f;
        const double fm = (2.0 * n / kappa) * f + fp;
        fp = f;
        f = fm;
        if (f > 1e250) {
          f *= 1e-250;
          fp *= 1e-250;
          sum *= 1e-250;
          for (int m = n; m <= kTerms; ++m) k.coef[m - 1] *= 1e-250;
        }
      }
      i0e = f / (f + 2.0 * sum);
      for (int n = 1; n <= kTerms; ++n) k.coef[n - 1] = (k.coef[n - 1] / f) / n;
    } else {
      double term = 1.0, s = 1.0;  // ive(0, x) ~ (2 pi x)^-1/2 sum_k ((2k-1)!!)^2 / (k! (8x)^k)
      for (int j = 1; j < 40; ++j) {
        const double t = term * ((2.0 * j - 1.0) * (2.0 * j - 1.0)) / (8.0 * kappa * j);
        if (!(t < term) || t < 1e-18 * s) break;
        term = t;
        s += t;
      }
      i0e = s / ::sqrt(2.0 * kPi * kappa);
    }
    k.log_i0e = flog(i0e);
    k.b = ::sqrt(2.0 / kPi) / i0e;
    double p = __builtin_rint(1.0 + 28.0 + 0.5 * kappa - 100.0 / (kappa + 5.0));  // helper.py:344
    k.p = p < 5.0 ? 5.0 : (p > 50.0 ? 50.0 : p);
    return k;
  }
  static M355_DEV double logpdf(double x, const K& k) {  // vonmises.py:45-46
    return (k.kappa * (::cos(x - k.mu) - 1.0) - flog(2.0 * kPi)) - k.log_i0e;
  }
  static M355_DEV double pdf(double x, const K& k) { return exp_(logpdf(x, k)); }  // vonmises.py:41-42
  static M355_DEV double cdf(double x, const K& k) {  // helper.py:337-372
    const double xc = x - k.mu;
    const double ix = __builtin_rint(xc / (2.0 * kPi));
    const double xw = xc - ix * 2.0 * kPi;
    double r;
    if (k.kappa < 50.0) {
      double s = 0.0;
      const int p = (int)k.p;
      for (int n = 1; n <= p; ++n) s += k.coef[n - 1] * ::sin(n * xw);
      r = (0.5 + xw / (2.0 * kPi)) + s / kPi;
      r = r < 0.0 ? 0.0 : (r > 1.0 ? 1.0 : r);
    } else {
      const double z = k.b * ::sin(xw / 2.0);
      r = 0.5 * (1.0 + erf_(z / kSqrt2));
    }
    return r + ix;
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return pdf(x, k);
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_LOGCDF) return flog(cdf(x, k));    // vonmises.py:57-58
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);        // vonmises.py:61-62
    if constexpr (FN == M355_LOGSF) return flog1p(-cdf(x, k));  // vonmises.py:65-66
    if constexpr (FN == M355_PPF || FN == M355_ISF) {          // vonmises.py:73-74, 69-70
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return find_ppf(q, k.mu, -kInf, kInf, [&](double v) { return cdf(v, k); }, [&](double v) { return pdf(v, k); });
    }
    return kNaN;
  }
};

// ----------------------------------------------------------------------------------------- rice
// Exponentially scaled modified Bessel functions I_0, I_1 (pt.ive): ascending series below 20 (all terms
// positive), Hankel's asymptotic series above (terms fall to e^{-2x} < 1e-17 before they grow).
M355_DEV_HEAVY double ive01_(int order, double x) {
  if (!(x == x)) return kNaN;
  const double ax = __builtin_fabs(x);
  double r;
  if (ax < 20.0) {
    const double q = 0.25 * ax * ax;
    double term = (order == 0) ? 1.0 : 0.5 * ax, s = term;
    for (int k = 1; k < 200; ++k) {
      term *= q / ((double)k * (double)(k + order));
      s += term;
      if (term < 1e-17 * s) break;
    }
    r = s * exp_(-ax);
  } else {
    const double mu = (order == 0) ? 0.0 : 4.0;
    double term = 1.0, s = 1.0;
    for (int k = 1; k < 60; ++k) {
      const double f = (2.0 * k - 1.0);
      const double t = -term * (mu - f * f) / (8.0 * ax * k);
      if (!(__builtin_fabs(t) < __builtin_fabs(term))) break;
      term = t;
      s += t;
      if (__builtin_fabs(t) < 1e-17 * __builtin_fabs(s)) break;
    }
    r = s / sqrt_(2.0 * kPi * ax);
  }
  return (order == 1 && x < 0.0) ? -r : r;
}

// ncx2_cdf(x, df = 2, nc) of helper.py:449-512: sum_{j<800} w_j P(j + 1, x/2), w_j = Poisson(j; nc/2) (clipped in
// log space to +-700).  With t_i = e^{-hx} hx^i / i! the regularised P(j + 1, hx) is sum_{i > j} t_i, so the
// double sum is sum_{i >= 1} t_i W_{min(i-1, 799)} with the running W_m = sum_{j <= m} w_j: one forward pass,
// every term positive (no cancellation for small x).
M355_DEV_HEAVY double ncx2_cdf_df2(double x, double nc) {
  if (!(x == x) || !(nc == nc)) return kNaN;
  if (!(nc < 1000.0)) {  // normal approximation, helper.py:505-509
    const double sd = sqrt_(2.0 * (2.0 + 2.0 * nc));
    return 0.5 * (1.0 + erf_((x - (2.0 + nc)) / (sd * kSqrt2)));
  }
  const double hnc = 0.5 * nc, hx = 0.5 * x;
  if (!(hnc > 0.0)) return kNaN;  // 0 * log(0) in the reference's log-weights
  if (!(hx > 0.0)) return hx == 0.0 ? 0.0 : kNaN;
  const double kMinW = 9.8596765437597708e-305;  // exp(-700): the reference's clip
  // Poisson weights forward from w_0 = exp(-hnc) (hnc < 500: no underflow)
  double w = exp_(-hnc);
  double W = w < kMinW ? kMinW : w;  // W_0
  int j = 0;                         // W holds W_j
  if (hx == kInf) return kNaN;
  // unnormalised Poisson terms u_i = c hx^i / i! by forward recurrence from i0 (terms below it are < e^-700 of the
  // mode), normalised at the end by Z = sum_i u_i: the rounding drift of the recurrence is common to S and Z
  long long i0 = 0;
  if (hx > 600.0) {
    const double lo = hx - 40.0 * sqrt_(hx);
    if (lo > 1.0) i0 = (long long)lo;
  }
  auto advance_W = [&](long long upto) {  // make W = W_{min(upto, 799)}
    const int target = (int)(upto < 799 ? upto : 799);
    while (j < target) {
      ++j;
      w *= hnc / (double)j;
      W += (w < kMinW ? kMinW : w);
    }
  };
  double u = 1.0, S = 0.0, Z = 0.0;
  const long long imax = i0 + 8000;
  for (long long i = i0; i < imax; ++i) {
    Z += u;
    if (i >= 1) {
      advance_W(i - 1);
      S += u * W;
    }
    u *= hx / (double)(i + 1);
    if (u > 1e200) {
      u *= 1e-200;
      S *= 1e-200;
      Z *= 1e-200;
    }
    if ((double)i > hx && u < 1e-19 * Z) break;
  }
  S = S / Z;
  return S;
}

template <> struct Dist<M355_RICE> {
  static constexpr int NP = 2;
  static constexpr bool kDeviceDerive = true;  // the Bessel series of the mean use the device exp
  struct K { double nu, sigma, sigma2, c, a, nc, mean; };  // c = -2 log sigma - nu^2/(2 sigma^2); a = nu/sigma
  static M355_DEV K derive(const double* p) {
    K k;
    k.nu = p[0];
    k.sigma = p[1];
    k.sigma2 = p[1] * p[1];
    k.c = -2.0 * flog(p[1]);
    k.a = p[0] / p[1];
    k.nc = k.a * k.a;
    const double q = (p[0] * p[0]) / (2.0 * k.sigma2);  // rice.py:13-28
    k.mean = p[1] * sqrt_(kPi / 2.0) * ((1.0 + q) * ive01_(0, q / 2.0) + q * ive01_(1, q / 2.0));
    return k;
  }
  static M355_DEV double logpdf(double x, const K& k) {  // rice.py:65-78
    if (x <= 0.0) return -kInf;
    if (x == kInf) return kNaN;
    const double z = x * k.nu / k.sigma2;
    return (((flog(x) + k.c) - (x * x + k.nu * k.nu) / (2.0 * k.sigma2)) + flog(ive01_(0, z))) + z;
  }
  static M355_DEV double pdf(double x, const K& k) { return exp_(logpdf(x, k)); }  // rice.py:61-62
  static M355_DEV double cdf(double x, const K& k) {  // rice.py:81-83, helper.py:515-539
    const double b = x / k.sigma;
    const double prob = (b <= 0.0) ? 0.0 : ncx2_cdf_df2(b * b, k.nc);
    return cdf_bounds(prob, x, 0.0, kInf);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return pdf(x, k);
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_LOGCDF) return (x <= 0.0) ? -kInf : flog(cdf(x, k));  // rice.py:86-91
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);                            // rice.py:94-95
    if constexpr (FN == M355_LOGSF) return flog(1.0 - cdf(x, k));                   // rice.py:98-99
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                              // rice.py:102-107
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return find_ppf(q, k.mean, 0.0, kInf, [&](double v) { return cdf(v, k); }, [&](double v) { return pdf(v, k); });
    }
    return kNaN;
  }
};

// ----------------------------------------------------------------------------------- polyagamma
// Density: the 20-term alternating Jacobi series of polyagamma.py:31-68 (log terms shifted by their maximum,
// signed sum in linear space, floored at 1e-300) times the exponential tilt (:113-117); the parameter-only part of
// each log term (gammaln(n + h) - gammaln(n + 1) + log(2n + h) - log sqrt(2 pi)) sits in K.  cdf: the reference's
// 500-point trapezoid of that density from 1e-10 to x (:124-134), evaluated per element (10^4 exp per cdf).  ppf:
// its own clipped Newton iteration from a log-normal start (:149-184), 50 steps at most, per lane with early exit.
template <> struct Dist<M355_POLYAGAMMA> {
  static constexpr int NP = 2;
  static constexpr int kN = 20;
  static constexpr double kLower = 1e-10;  // polyagamma.py:15-16
  struct K {
    double h, z, tilt, hz2, pre, mu_ln, sigma_ln;  // tilt = h log cosh(z/2); hz2 = z^2/2; pre = (h-1) log 2 - gammaln(h)
    double A[kN];                                  // parameter-only part of the n-th log term
    double c8[kN];                                 // (2n + h)^2 / 8
  };
  static M355_HD K derive(const double* p) {
    K k;
    const double h = p[0], z = p[1];
    k.h = h;
    k.z = z;
    const double a = ::fabs(z / 2.0);
    k.tilt = h * ((a + ::log1p(::exp(-2.0 * a))) - kLog2);  // polyagamma.py:25-28
    k.hz2 = (z * z) / 2.0;
    k.pre = (h - 1.0) * kLog2 - lgamma_(h);
    for (int n = 0; n < kN; ++n) {
      const double c = 2.0 * n + h;
      k.A[n] = ((lgamma_(n + h) - lgamma_(n + 1.0)) + flog(c)) - kHalfLog2Pi;
      k.c8[n] = (c * c) / 8.0;
    }
    const bool small = ::fabs(z) < 1e-6;  // polyagamma.py:71-77, 88-94
    const double sz = small ? 1.0 : z;
    const double ch = ::cosh(sz / 2.0);
    const double m = small ? h / 4.0 : h / (2.0 * sz) * ::tanh(sz / 2.0);
    const double v = small ? h / 24.0 : h / (4.0 * (sz * sz * sz)) * (::sinh(sz) - sz) / (ch * ch);
    const double s2 = ::log1p(v / (m * m));
    k.mu_ln = flog(m) - s2 / 2.0;
    k.sigma_ln = ::sqrt(s2);
    return k;
  }
  static M355_DEV double logpdf(double x, const K& k) {  // polyagamma.py:113-117
    if (x <= 0.0) return -kInf;
    const double lx15 = 1.5 * flog(x);
    const double inv = 1.0 / x;
    double l[kN], mx = -kInf;
    bool nan = false;
#pragma unroll
    for (int n = 0; n < kN; ++n) {
      l[n] = (k.A[n] - lx15) - k.c8[n] * inv;
      nan = nan || !(l[n] == l[n]);
      mx = l[n] > mx ? l[n] : mx;
    }
    if (nan) mx = kNaN;  // np.max propagates NaN
    double ssum = 0.0;
#pragma unroll
    for (int n = 0; n < kN; ++n) {
      const double e = exp_(l[n] - mx);
      ssum += (n & 1) ? -e : e;
    }
    const double fl = ssum > 1e-300 ? ssum : ((ssum == ssum) ? 1e-300 : ssum);
    return (k.tilt - k.hz2 * x) + (k.pre + (flog(fl) + mx));
  }
  static M355_DEV double pdf(double x, const K& k) { return exp_(logpdf(x, k)); }  // polyagamma.py:120-121
  M355_DEV_HEAVY double cdf(double x, const K& k) {  // polyagamma.py:124-134
    if (x < 0.0) return 0.0;       // cdf_bounds (helper.py:4-26)
    if (x >= kInf) return 1.0;
    if (!(x == x)) return kNaN;
    const double step = (x - kLower) / 499.0;
    double s = 0.0;
    for (int i = 1; i < 499; ++i) s += pdf(kLower + (double)i * step, k);
    return step * ((0.5 * pdf(kLower, k) + s) + 0.5 * pdf(kLower + 499.0 * step, k));
  }
  M355_DEV_HEAVY double ppf(double q, const K& k) {  // polyagamma.py:149-184
    double x = exp_(k.mu_ln + k.sigma_ln * kSqrt2 * erfinv_(2.0 * q - 1.0));
    x = x > kLower ? x : ((x == x) ? kLower : x);
    for (int it = 0; it < 50; ++it) {
      const double c = cdf(x, k);
      const double f0 = pdf(x, k);
      const double f = f0 > 1e-10 ? f0 : 1e-10;
      double delta = (c - q) / ((f0 == f0) ? f : f0);
      const double ax = __builtin_fabs(x);
      const double ms = ax > 0.5 ? ax : ((ax == ax) ? 0.5 : ax);
      delta = delta < -ms ? -ms : (delta > ms ? ms : delta);
      double xn = x - delta;
      xn = xn > kLower ? xn : ((xn == xn) ? kLower : xn);
      if (__builtin_fabs(xn - x) < 1e-8) break;
      x = xn;
      if (!(x == x)) break;
    }
    return ppf_bounds_cont(x, q, 0.0, kInf);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return pdf(x, k);
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_LOGCDF) return (x <= 0.0) ? -kInf : flog(cdf(x, k));  // polyagamma.py:137-138
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);                            // polyagamma.py:141-142
    if constexpr (FN == M355_LOGSF) return flog1p(-cdf(x, k));                      // polyagamma.py:145-146
    if constexpr (FN == M355_PPF) return ppf(x, k);
    if constexpr (FN == M355_ISF) return ppf(1.0 - x, k);                           // polyagamma.py:187-188
    return kNaN;
  }
};

}  // namespace m355
