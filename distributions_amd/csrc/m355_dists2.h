// Scalar bodies of the closed-form and re-parameterised continuous families:
// the same Dist<D>::derive / eval<FN> template as m355_dists.h, one struct per reference module.
// Every formula follows the reference line by line (file:line cited per function; paths relative to
// /root/reference/pytensor_distributions); pt.switch is a select, so only the selected arm's value
// is produced.  Parameter-only logarithms are hoisted into derive().
#pragma once
#include "m355_dists.h"

namespace m355 {

constexpr double kLog2 = 0.69314718055994530942;
constexpr double kLogHalf = -0.69314718055994530942;
constexpr double kHalfLog2Pi = 0.91893853320467274178;  // 0.5 * log(2 pi)

// ---------------------------------------------------------------------------------- exponential
template <> struct Dist<M355_EXPONENTIAL> {
  static constexpr int NP = 1;
  struct K { double lam, loglam; };
  static M355_HD K derive(const double* p) { return K{p[0], flog(p[0])}; }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return (x < 0.0) ? -kInf : k.loglam - k.lam * x;          // exponential.py:71-72
    if constexpr (FN == M355_PDF) return (x < 0.0) ? 0.0 : k.lam * exp_(-k.lam * x);           // exponential.py:50-51
    if constexpr (FN == M355_CDF) return cdf_bounds(1.0 - exp_(-k.lam * x), x, 0.0, kInf);     // exponential.py:41-43
    if constexpr (FN == M355_SF) return (x < 0.0) ? 1.0 : exp_(-k.lam * x);                    // exponential.py:59-60
    if constexpr (FN == M355_LOGCDF) return (x < 0.0) ? -kInf : flog1p(-exp_(-k.lam * x));     // exponential.py:67-68
    if constexpr (FN == M355_LOGSF) return (x < 0.0) ? 0.0 : -k.lam * x;                       // exponential.py:75-76
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                                          // exponential.py:54-56, 46-47
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(-flog(1.0 - q) / k.lam, q, 0.0, kInf);
    }
    return kNaN;
  }
};

// ------------------------------------------------------------------------------------ lognormal
template <> struct Dist<M355_LOGNORMAL> {
  static constexpr int NP = 2;
  struct K { double mu, sigma, logsigma; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], flog(p[1])}; }
  static M355_DEV double logpdf(double x, const K& k) {  // lognormal.py:83-91
    const double lx = flog(x);
    const double z = (lx - k.mu) / k.sigma;
    const double r = -lx - k.logsigma - kHalfLog2Pi - 0.5 * (z * z);
    return (x <= 0.0) ? -kInf : r;
  }
  static M355_DEV double logcdf(double x, const K& k) {  // lognormal.py:70-80
    if (x <= 0.0) return -kInf;
    return normal_logcdf_z((flog(x) - k.mu) / k.sigma);
  }
  static M355_DEV double logsf(double x, const K& k) {  // lognormal.py:94-104
    if (x <= 0.0) return 0.0;
    const double z = (flog(x) - k.mu) / k.sigma;
    if (z > 1.0) return flog(0.5 * erfcx_(z * kInvSqrt2)) - 0.5 * z * z;
    return flog1p(-0.5 * (1.0 + erf_(z * kInvSqrt2)));
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));  // lognormal.py:54-55
    if constexpr (FN == M355_CDF)                            // lognormal.py:45-47
      return cdf_bounds(0.5 * (1.0 + erf_((flog(x) - k.mu) / (k.sigma * kSqrt2))), x, 0.0, kInf);
    if constexpr (FN == M355_LOGCDF) return logcdf(x, k);
    if constexpr (FN == M355_LOGSF) return logsf(x, k);
    if constexpr (FN == M355_SF) return exp_(logsf(x, k));    // lognormal.py:62-63
    if constexpr (FN == M355_PPF || FN == M355_ISF) {        // lognormal.py:58-59, 50-51
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(exp_(k.mu + k.sigma * kSqrt2 * erfinv_(2.0 * q - 1.0)), q, 0.0, kInf);
    }
    return kNaN;
  }
};

// ----------------------------------------------------------------------------------- halfnormal
M355_DEV double halfnormal_logpdf(double x, double sigma, double logsigma) {  // halfnormal.py:67-70
  const double z = x / sigma;
  return (x < 0.0) ? -kInf : (-0.22579135264472743236 - logsigma) - 0.5 * (z * z);  // log(sqrt(2/pi))
}
M355_DEV double halfnormal_cdf(double x, double sigma) {  // halfnormal.py:38-39
  return (x < 0.0) ? 0.0 : erf_(x / (sigma * kSqrt2));
}
template <> struct Dist<M355_HALFNORMAL> {
  static constexpr int NP = 1;
  struct K { double sigma, logsigma; };
  static M355_HD K derive(const double* p) { return K{p[0], flog(p[0])}; }
  static M355_DEV double logcdf(double x, const K& k) {  // halfnormal.py:42-43
    return (x < 0.0) ? -kInf : flog(erf_(x / (k.sigma * kSqrt2)));
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return halfnormal_logpdf(x, k.sigma, k.logsigma);
    if constexpr (FN == M355_PDF) return exp_(halfnormal_logpdf(x, k.sigma, k.logsigma));  // halfnormal.py:50-51
    if constexpr (FN == M355_CDF) return halfnormal_cdf(x, k.sigma);
    if constexpr (FN == M355_LOGCDF) return logcdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - halfnormal_cdf(x, k.sigma);  // halfnormal.py:59-60
    if constexpr (FN == M355_LOGSF) return log1mexp_(logcdf(x, k));         // halfnormal.py:73-74
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                       // halfnormal.py:54-56, 46-47
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(k.sigma * kSqrt2 * erfinv_(q), q, 0.0, kInf);
    }
    return kNaN;
  }
};

// --------------------------------------------------------------------------------------- cauchy
M355_DEV double clip01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
template <> struct Dist<M355_CAUCHY> {
  static constexpr int NP = 2;
  struct K { double alpha, beta, c; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], -kLogPi - flog(p[1])}; }
  static M355_DEV double cdf(double x, const K& k) {  // cauchy.py:44-45
    return (1.0 / kPi) * atan_((x - k.alpha) / k.beta) + 0.5;
  }
  // sum of logpdf with the logarithms carried as a product (m355_math.h LogProd; protocol of Dist<M355_GAMMA>::SumAcc)
  struct SumAcc {
    LogProd px;
    double lin = 0.0, exc = 0.0;
    int cnt = 0;
  };
  static M355_DEV void sum_renorm(SumAcc& s) { s.px.renorm(); }
  static M355_DEV bool sum_ok(const K&) { return true; }
  static M355_DEV void sum_add(SumAcc& s, double x, const K& k) {
    const double z = (x - k.alpha) / k.beta;
    const double u = 1.0 + z * z;
    if (u < 0x1p1000) {
      s.px.mul(u);
      ++s.cnt;
    } else {
      s.exc += logpdf(x, k);
    }
  }
  static M355_DEV double sum_finish(const SumAcc& s, const K& k) { return ((double)s.cnt * k.c - s.px.log()) + s.exc; }
  static M355_DEV double logpdf(double x, const K& k) {  // cauchy.py:72-73
    const double z = (x - k.alpha) / k.beta;
    return k.c - flog(1.0 + z * z);
  }
  static M355_DEV double logsf(double x, const K& k) {  // cauchy.py:76-78
    return flog(clip01(0.5 - (1.0 / kPi) * atan_((x - k.alpha) / k.beta)));
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));        // cauchy.py:52-53
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_LOGCDF) return flog(clip01(cdf(x, k)));  // cauchy.py:68-69
    if constexpr (FN == M355_LOGSF) return logsf(x, k);
    if constexpr (FN == M355_SF) return exp_(logsf(x, k));          // cauchy.py:60-61
    if constexpr (FN == M355_PPF || FN == M355_ISF) {              // cauchy.py:56-57, 48-49
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(k.alpha + k.beta * tan_(kPi * (q - 0.5)), q, -kInf, kInf);
    }
    return kNaN;
  }
};

// ----------------------------------------------------------------------------------- halfcauchy
template <> struct Dist<M355_HALFCAUCHY> {
  static constexpr int NP = 1;
  struct K { double beta, c; };
  static M355_HD K derive(const double* p) { return K{p[0], kLog2 - flog(kPi * p[0])}; }
  static M355_DEV double logcdf(double x, const K& k) {  // halfcauchy.py:42-47
    return (x < 0.0) ? -kInf : flog(2.0 * atan_(x / k.beta) / kPi);
  }
  // sum of logpdf with the logarithms carried as a product (m355_math.h LogProd; protocol of Dist<M355_GAMMA>::SumAcc)
  struct SumAcc {
    LogProd px;
    double lin = 0.0, exc = 0.0;
    int cnt = 0;
  };
  static M355_DEV void sum_renorm(SumAcc& s) { s.px.renorm(); }
  static M355_DEV bool sum_ok(const K&) { return true; }
  static M355_DEV void sum_add(SumAcc& s, double x, const K& k) {
    const double z = x / k.beta;
    const double u = 1.0 + z * z;
    if ((x >= 0.0) & (u < 0x1p1000)) {
      s.px.mul(u);
      ++s.cnt;
    } else {
      s.exc += logpdf(x, k);
    }
  }
  static M355_DEV double sum_finish(const SumAcc& s, const K& k) { return ((double)s.cnt * k.c - s.px.log()) + s.exc; }
  static M355_DEV double logpdf(double x, const K& k) {  // halfcauchy.py:72-75
    const double z = x / k.beta;
    return (x < 0.0) ? -kInf : k.c - flog(1.0 + z * z);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));         // halfcauchy.py:54-55
    if constexpr (FN == M355_LOGCDF) return logcdf(x, k);
    if constexpr (FN == M355_CDF) return exp_(logcdf(x, k));         // halfcauchy.py:38-39
    if constexpr (FN == M355_SF) return 1.0 - exp_(logcdf(x, k));    // halfcauchy.py:63-64
    if constexpr (FN == M355_LOGSF) return log1mexp_(logcdf(x, k));  // halfcauchy.py:78-79
    if constexpr (FN == M355_PPF || FN == M355_ISF) {               // halfcauchy.py:58-60, 50-51
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(k.beta * tan_(kPi / 2.0 * q), q, 0.0, kInf);
    }
    return kNaN;
  }
};

// -------------------------------------------------------------------------------------- laplace
template <> struct Dist<M355_LAPLACE> {
  static constexpr int NP = 2;
  struct K { double mu, b, logb; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], flog(p[1])}; }
  static M355_DEV double cdf(double x, const K& k) {  // laplace.py:50-52
    const double r = (x < k.mu) ? 0.5 * exp_((x - k.mu) / k.b) : 1.0 - 0.5 * exp_(-(x - k.mu) / k.b);
    return cdf_bounds(r, x, -kInf, kInf);
  }
  static M355_DEV double logcdf_raw(double x, double mu, double b) {  // laplace.py:76-86
    const double y = (x - mu) / b;
    if (y <= 0.0) return kLogHalf + y;
    if (y > 1.0) return flog1p(-0.5 * exp_(-y));
    return flog(1.0 - 0.5 * exp_(-y));  // (NaN y lands here, as in the reference's nested switch)
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return (kLogHalf - __builtin_fabs((x - k.mu) / k.b)) - k.logb;  // laplace.py:72-73
    if constexpr (FN == M355_PDF) return 0.5 / k.b * exp_(-__builtin_fabs(x - k.mu) / k.b);          // laplace.py:46-47
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);                                              // laplace.py:55-56
    if constexpr (FN == M355_LOGCDF) return logcdf_raw(x, k.mu, k.b);
    if constexpr (FN == M355_LOGSF) return logcdf_raw(-x, -k.mu, k.b);                                // laplace.py:89-90
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                                                 // laplace.py:59-61, 64-65
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      const double r = (q < 0.5) ? k.mu + k.b * flog(2.0 * q) : k.mu - k.b * flog(2.0 * (1.0 - q));
      return ppf_bounds_cont(r, q, -kInf, kInf);
    }
    return kNaN;
  }
};

// ------------------------------------------------------------------------------------- logistic
template <> struct Dist<M355_LOGISTIC> {
  static constexpr int NP = 2;
  static constexpr bool kLogTable = true;  // logpdf uses flog1p_tab
  struct K { double mu, s, logs; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], flog(p[1])}; }
  static M355_DEV double logpdf(double x, const K& k) {  // logistic.py:74-76
    const double z = (x - k.mu) / k.s;
    return (x == -kInf) ? -kInf : (-k.logs - z) - 2.0 * flog1p_tab(exp_(-z));
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));                           // logistic.py:54-55
    if constexpr (FN == M355_CDF) return 1.0 / (1.0 + exp_(-(x - k.mu) / k.s));        // logistic.py:46-47
    if constexpr (FN == M355_LOGCDF) return -log1pexp_(-(x - k.mu) / k.s);             // logistic.py:70-71
    if constexpr (FN == M355_LOGSF) return -log1pexp_((x - k.mu) / k.s);               // logistic.py:79-80
    if constexpr (FN == M355_SF) return exp_(-log1pexp_((x - k.mu) / k.s));            // logistic.py:62-63
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                                  // logistic.py:58-59, 50-51
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(k.mu + k.s * flog(q / (1.0 - q)), q, -kInf, kInf);
    }
    return kNaN;
  }
};

// --------------------------------------------------------------------------------------- gumbel
template <> struct Dist<M355_GUMBEL> {
  static constexpr int NP = 2;
  struct K { double mu, beta, logbeta; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], flog(p[1])}; }
  static M355_DEV double logsf(double x, const K& k) {  // gumbel.py:79-80
    return flog1p(-exp_(-exp_(-(x - k.mu) / k.beta)));
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) {  // gumbel.py:74-76
      const double z = (x - k.mu) / k.beta;
      return (-z - exp_(-z)) - k.logbeta;
    }
    if constexpr (FN == M355_PDF) {  // gumbel.py:53-54
      const double z = (x - k.mu) / k.beta;
      return exp_((-z - exp_(-z)) - k.logbeta);
    }
    if constexpr (FN == M355_CDF) return cdf_bounds(exp_(-exp_(-(x - k.mu) / k.beta)), x, -kInf, kInf);  // gumbel.py:44-46
    if constexpr (FN == M355_LOGCDF) return -exp_(-(x - k.mu) / k.beta);                                 // gumbel.py:70-71
    if constexpr (FN == M355_LOGSF) return logsf(x, k);
    if constexpr (FN == M355_SF) return exp_(logsf(x, k));                                               // gumbel.py:62-63
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                                                    // gumbel.py:57-59, 49-50
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(k.mu - k.beta * flog(-flog(q)), q, -kInf, kInf);
    }
    return kNaN;
  }
};

// -------------------------------------------------------------------------------------- weibull
template <> struct Dist<M355_WEIBULL> {
  static constexpr int NP = 2;
  static constexpr bool kLogTable = true;  // logpdf: one table log shared by log(z) and z**alpha
  struct K { double alpha, beta, c; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], flog(p[0] / p[1])}; }
  static M355_DEV double logpdf(double x, const K& k) {  // weibull.py:82-87
    if (x < 0.0) return -kInf;
    const double z = x / k.beta;
    const double lz = flog_tab(z);
    return (k.c + xlogy0_l(k.alpha - 1.0, lz)) - pow_l(z, lz, k.alpha);
  }
  static M355_DEV double logcdf(double x, const K& k) {  // weibull.py:72-79
    return (x < 0.0) ? -kInf : log1mexp_(-pow_(x / k.beta, k.alpha));
  }
  static M355_DEV double logsf(double x, const K& k) {  // weibull.py:90-91
    return (x < 0.0) ? 0.0 : -pow_(x / k.beta, k.alpha);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));  // weibull.py:55-56
    if constexpr (FN == M355_LOGCDF) return logcdf(x, k);
    if constexpr (FN == M355_CDF) return exp_(logcdf(x, k));  // weibull.py:47-48
    if constexpr (FN == M355_LOGSF) return logsf(x, k);
    if constexpr (FN == M355_SF) return exp_(logsf(x, k));    // weibull.py:64-65
    if constexpr (FN == M355_PPF || FN == M355_ISF) {        // weibull.py:59-61, 51-52
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(k.beta * pow_(-flog(1.0 - q), 1.0 / k.alpha), q, 0.0, kInf);
    }
    return kNaN;
  }
};

// --------------------------------------------------------------------------------------- pareto
template <> struct Dist<M355_PARETO> {
  static constexpr int NP = 2;
  struct K { double alpha, m, c; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], flog(p[0]) + p[0] * flog(p[1])}; }
  static M355_DEV double cdf(double x, const K& k) {  // pareto.py:47-48
    return (x < k.m) ? 0.0 : 1.0 - pow_(k.m / x, k.alpha);
  }
  // sum of logpdf with the logarithms carried as a product (m355_math.h LogProd; protocol of Dist<M355_GAMMA>::SumAcc)
  struct SumAcc {
    LogProd px;
    double lin = 0.0, exc = 0.0;
    int cnt = 0;
  };
  static M355_DEV void sum_renorm(SumAcc& s) { s.px.renorm(); }
  static M355_DEV bool sum_ok(const K& k) { return __builtin_fabs(k.alpha + 1.0) <= kSumAccMaxCoef; }
  static M355_DEV void sum_add(SumAcc& s, double x, const K& k) {
    if ((x >= k.m) & (x >= kMinNormal) & (x < kInf)) {
      s.px.mul(x);
      ++s.cnt;
    } else {
      s.exc += eval<M355_LOGPDF>(x, k);
    }
  }
  static M355_DEV double sum_finish(const SumAcc& s, const K& k) {
    return ((double)s.cnt * k.c - (k.alpha + 1.0) * s.px.log()) + s.exc;
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return (x < k.m) ? -kInf : k.c - (k.alpha + 1.0) * flog(x);  // pareto.py:77-80
    if constexpr (FN == M355_PDF) return exp_((x < k.m) ? -kInf : k.c - (k.alpha + 1.0) * flog(x));  // pareto.py:51-52
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);                                          // pareto.py:60-61
    if constexpr (FN == M355_LOGCDF) return (x < k.m) ? -kInf : flog(1.0 - pow_(k.m / x, k.alpha));  // pareto.py:73-74
    if constexpr (FN == M355_LOGSF) return flog(1.0 - cdf(x, k));                                 // pareto.py:83-84
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                                             // pareto.py:55-57, 64-65
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(k.m * pow_(1.0 - q, -1.0 / k.alpha), q, k.m, kInf);
    }
    return kNaN;
  }
};

// -------------------------------------------------------------------------------------- uniform
template <> struct Dist<M355_UNIFORM> {
  static constexpr int NP = 2;
  struct K { double lo, hi, w, logh; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], p[1] - p[0], -flog(p[1] - p[0])}; }
  static M355_DEV double cdf(double x, const K& k) {  // uniform.py:41-46
    if (x < k.lo) return 0.0;
    if (x > k.hi) return 1.0;
    return (x - k.lo) / k.w;
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return ((x < k.lo) | (x > k.hi)) ? -kInf : k.logh;          // uniform.py:78-80
    if constexpr (FN == M355_PDF) return exp_(((x < k.lo) | (x > k.hi)) ? -kInf : k.logh);       // uniform.py:49-50
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);                                         // uniform.py:58-59
    if constexpr (FN == M355_LOGCDF) {                                                           // uniform.py:70-75
      if (x < k.lo) return -kInf;
      if (x > k.hi) return 0.0;
      return flog((x - k.lo) / k.w);
    }
    if constexpr (FN == M355_LOGSF) return flog(1.0 - cdf(x, k));                                // uniform.py:83-84
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                                            // uniform.py:53-55, 62-63
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(k.lo + q * k.w, q, k.lo, k.hi);
    }
    return kNaN;
  }
};

// ----------------------------------------------------------------------------------- triangular
template <> struct Dist<M355_TRIANGULAR> {
  static constexpr int NP = 3;
  struct K { double lo, c, hi, dl, du; };  // dl = (hi-lo)(c-lo), du = (hi-lo)(hi-c)
  static M355_HD K derive(const double* p) {
    return K{p[0], p[1], p[2], (p[2] - p[0]) * (p[1] - p[0]), (p[2] - p[0]) * (p[2] - p[1])};
  }
  static M355_DEV double logsf(double x, const K& k) {  // triangular.py:116-129
    if (x >= k.hi) return -kInf;
    if (x >= k.c) return flog(((k.hi - x) * (k.hi - x)) / k.du);
    if (x > k.lo) return flog1p(-((x - k.lo) * (x - k.lo)) / k.dl);
    return 0.0;
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_PDF) {  // triangular.py:44-50
      const double r = (x < k.c) ? 2.0 * (x - k.lo) / k.dl : 2.0 * (k.hi - x) / k.du;
      return ((k.lo <= x) & (x <= k.hi)) ? r : 0.0;
    }
    if constexpr (FN == M355_LOGPDF) {  // triangular.py:91-97
      const double r = (x < k.c) ? flog(2.0 * (x - k.lo) / k.dl) : flog(2.0 * (k.hi - x) / k.du);
      return ((k.lo <= x) & (x <= k.hi)) ? r : -kInf;
    }
    if constexpr (FN == M355_CDF) {  // triangular.py:53-67
      if (x <= k.lo) return 0.0;
      if (x <= k.c) return ((x - k.lo) * (x - k.lo)) / k.dl;
      if (x < k.hi) return 1.0 - ((k.hi - x) * (k.hi - x)) / k.du;
      return 1.0;
    }
    if constexpr (FN == M355_LOGCDF) {  // triangular.py:100-113
      if (x <= k.lo) return -kInf;
      if (x <= k.c) return flog(((x - k.lo) * (x - k.lo)) / k.dl);
      if (x < k.hi) return flog1p(-((k.hi - x) * (k.hi - x)) / k.du);
      return 0.0;
    }
    if constexpr (FN == M355_LOGSF) return logsf(x, k);
    if constexpr (FN == M355_SF) return exp_(logsf(x, k));  // triangular.py:78-79
    if constexpr (FN == M355_PPF || FN == M355_ISF) {      // triangular.py:70-75, 82-83 (no bound wrapper in the reference)
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return (q < (k.c - k.lo) / (k.hi - k.lo)) ? k.lo + sqrt_(k.dl * q) : k.hi - sqrt_(k.du * (1.0 - q));
    }
    return kNaN;
  }
};

// ---------------------------------------------------------------------------------- kumaraswamy
template <> struct Dist<M355_KUMARASWAMY> {
  static constexpr int NP = 2;
  static constexpr bool kLogTable = true;  // logpdf: table logs, log(x) shared with x**a
  struct K { double a, b, logab; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], flog(p[0] * p[1])}; }
  static M355_DEV double logpdf(double x, const K& k) {  // kumaraswamy.py:80-85
    if ((x <= 0.0) | (x >= 1.0)) return -kInf;
    const double lx = flog_tab(x);
    return (k.logab + xlogy0_l(k.a - 1.0, lx)) + xlogy0_l(k.b - 1.0, flog_tab(1.0 - pow_l(x, lx, k.a)));
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));  // kumaraswamy.py:58-59
    if constexpr (FN == M355_CDF) return cdf_bounds(1.0 - pow_(1.0 - pow_(x, k.a), k.b), x, 0.0, 1.0);  // kumaraswamy.py:53-55
    if constexpr (FN == M355_SF) return sf_bounds(pow_(1.0 - pow_(x, k.a), k.b), x, 0.0, 1.0);          // kumaraswamy.py:67-68
    if constexpr (FN == M355_LOGSF) {                                                                  // kumaraswamy.py:88-89
      if (x <= 0.0) return 0.0;
      if (x >= 1.0) return -kInf;
      return k.b * flog(1.0 - pow_(x, k.a));
    }
    if constexpr (FN == M355_LOGCDF) {  // kumaraswamy.py:92-93
      if (x <= 0.0) return -kInf;
      if (x >= 1.0) return 0.0;
      return flog(1.0 - pow_(1.0 - pow_(x, k.a), k.b));
    }
    if constexpr (FN == M355_PPF || FN == M355_ISF) {  // kumaraswamy.py:62-64, 71-72
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(pow_(1.0 - pow_(1.0 - q, 1.0 / k.b), 1.0 / k.a), q, 0.0, 1.0);
    }
    return kNaN;
  }
};

// ---------------------------------------------------------------------------------------- moyal
template <> struct Dist<M355_MOYAL> {
  static constexpr int NP = 2;
  struct K { double mu, sigma, c; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], -flog(p[1]) - kHalfLog2Pi}; }
  static M355_DEV double cdf(double x, const K& k) {  // moyal.py:43-45
    const double z = (x - k.mu) / k.sigma;
    return 1.0 - erf_(exp_(-z / 2.0) * kInvSqrt2);
  }
  static M355_DEV double logpdf(double x, const K& k) {  // moyal.py:75-77
    const double z = (x - k.mu) / k.sigma;
    return -0.5 * (z + exp_(-z)) + k.c;
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));          // moyal.py:48-49
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);              // moyal.py:57-58
    if constexpr (FN == M355_LOGCDF) return flog(cdf(x, k));          // moyal.py:70-72
    if constexpr (FN == M355_LOGSF) return flog(1.0 - cdf(x, k));     // moyal.py:80-81
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                // moyal.py:52-54, 61-62
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      const double e = erfinv_(1.0 - q);
      return ppf_bounds_cont(k.sigma * -flog(2.0 * (e * e)) + k.mu, q, -kInf, kInf);
    }
    return kNaN;
  }
};

// ---------------------------------------------------------------------------- asymmetriclaplace
template <> struct Dist<M355_ASYMMETRICLAPLACE> {
  static constexpr int NP = 3;
  struct K { double mu, b, kappa, kinv, kk, c; };  // kk = kappa + 1/kappa
  static M355_HD K derive(const double* p) {
    const double kinv = 1.0 / p[2];
    return K{p[0], p[1], p[2], kinv, p[2] + kinv, flog(p[2] + kinv), };
  }
  static M355_DEV double cdf(double x, const K& k) {  // asymmetriclaplace.py:43-52
    const double xn = (x - k.mu) / k.b;
    return (xn >= 0.0) ? 1.0 - exp_(-xn * k.kappa) * (k.kinv / k.kk) : exp_(xn * k.kinv) * (k.kappa / k.kk);
  }
  static M355_DEV double logpdf(double x, const K& k) {  // asymmetriclaplace.py:88-95
    const double xn = (x - k.mu) / k.b;
    const double a = (xn >= 0.0) ? xn * -k.kappa : xn * k.kinv;
    return (a - k.c) - flog(k.b);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));        // asymmetriclaplace.py:68-69
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);            // asymmetriclaplace.py:72-73
    if constexpr (FN == M355_LOGCDF) return flog(cdf(x, k));        // asymmetriclaplace.py:98-99
    if constexpr (FN == M355_LOGSF) return flog(1.0 - cdf(x, k));   // asymmetriclaplace.py:102-103
    if constexpr (FN == M355_PPF || FN == M355_ISF) {              // asymmetriclaplace.py:55-65, 76-77
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      const double v = (q >= k.kappa / k.kk) ? -flog((1.0 - q) * k.kk * k.kappa) * k.kinv : flog(q * k.kk / k.kappa) * k.kappa;
      return ppf_bounds_cont(v * k.b + k.mu, q, -kInf, kInf);
    }
    return kNaN;
  }
};

// ---------------------------------------------------------------------------------- loglogistic
template <> struct Dist<M355_LOGLOGISTIC> {
  static constexpr int NP = 2;
  static constexpr bool kLogTable = true;  // logpdf: table logs, log(z) shared with z**beta
  struct K { double alpha, beta, c; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], flog(p[1]) - flog(p[0])}; }
  static M355_DEV double logpdf(double x, const K& k) {  // loglogistic.py:70-73
    if (x <= 0.0) return -kInf;
    if (x == kInf) return -kInf;
    const double z = x / k.alpha;
    const double lz = flog_tab(z);
    return (k.c + (k.beta - 1.0) * lz) - 2.0 * flog1p_tab(pow_l(z, lz, k.beta));
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));  // loglogistic.py:66-67
    if constexpr (FN == M355_CDF) return cdf_bounds(1.0 / (1.0 + pow_(x / k.alpha, -k.beta)), x, 0.0, kInf);  // loglogistic.py:76-79
    if constexpr (FN == M355_LOGCDF) return (x <= 0.0) ? -kInf : -flog1p(pow_(x / k.alpha, -k.beta));         // loglogistic.py:82-88
    if constexpr (FN == M355_SF) return (x <= 0.0) ? 1.0 : 1.0 / (1.0 + pow_(x / k.alpha, k.beta));            // loglogistic.py:91-97
    if constexpr (FN == M355_LOGSF) return (x <= 0.0) ? 0.0 : -flog1p(pow_(x / k.alpha, k.beta));              // loglogistic.py:100-106
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                                                         // loglogistic.py:109-111, 114-115
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(k.alpha * pow_(q / (1.0 - q), 1.0 / k.beta), q, 0.0, kInf);
    }
    return kNaN;
  }
};

// ---------------------------------------------------------------------------------- logitnormal
template <> struct Dist<M355_LOGITNORMAL> {
  static constexpr int NP = 2;
  static constexpr bool kLogTable = true;  // logpdf uses flog_tab / flog1p_tab
  struct K { double mu, sigma, c; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], -flog(p[1]) - kHalfLog2Pi}; }
  static M355_DEV double logit(double x) { return flog(x) - flog1p(-x); }  // logitnormal.py:11-12
  static M355_DEV double logpdf(double x, const K& k) {                    // logitnormal.py:122-132
    if ((x <= 0.0) | (x >= 1.0)) return -kInf;
    const double lx = flog_tab(x), lc = flog1p_tab(-x);  // 0 < x < 1: 1 - x is exact above 1/2, corrected below
    const double z = ((lx - lc) - k.mu) / k.sigma;
    return ((-0.5 * (z * z) + k.c) - lx) - lc;
  }
  static M355_DEV double logcdf(double x, const K& k) {  // logitnormal.py:141-156
    if (x <= 0.0) return -kInf;
    if (x >= 1.0) return 0.0;
    return normal_logcdf_z((logit(x) - k.mu) / k.sigma);
  }
  static M355_DEV double logsf(double x, const K& k) {  // logitnormal.py:163-178
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return -kInf;
    const double z = (logit(x) - k.mu) / k.sigma;
    if (z > 1.0) return flog(0.5 * erfcx_(z * kInvSqrt2)) - 0.5 * z * z;
    return flog1p(-0.5 * (1.0 + erf_(z * kInvSqrt2)));
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));  // logitnormal.py:118-119
    if constexpr (FN == M355_CDF)                            // logitnormal.py:135-138
      return cdf_bounds(0.5 * (1.0 + erf_((logit(x) - k.mu) / (k.sigma * kSqrt2))), x, 0.0, 1.0);
    if constexpr (FN == M355_LOGCDF) return logcdf(x, k);
    if constexpr (FN == M355_LOGSF) return logsf(x, k);
    if constexpr (FN == M355_SF) return exp_(logsf(x, k));    // logitnormal.py:159-160
    if constexpr (FN == M355_PPF || FN == M355_ISF) {        // logitnormal.py:181-182, 185-186 (through normal.ppf, normal.py:58-59)
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      const double nq = ppf_bounds_cont(k.mu + k.sigma * kSqrt2 * erfinv_(2.0 * q - 1.0), q, -kInf, kInf);
      return ppf_bounds_cont(sigmoid_(nq), q, 0.0, 1.0);
    }
    return kNaN;
  }
};

// --------------------------------------------------------------------------------- inversegamma
template <> struct Dist<M355_INVERSEGAMMA> {
  static constexpr int kMinWaves = 3;  // binned kernel: 168-VGPR budget around the inlined gammaincinv_p (m355_elemwise.h)
  static constexpr int kBinned = 1;
  // quantile tables: the reference takes gammaincinv at 1 - q (inversegamma.py:60-61, scaledinversechisquared.py:56-59),
  // which rounds for small q -- a staircase in q, 3e-10 of x at q = 1e-8 -- but is the smooth TRUE quantile of
  // q' = 1 - fl(1 - q).  So the lookup forms its logit from (1 - den, den) with the reference's own den (tab_q_den),
  // and the nodes below the median are true quantiles: one Newton step on logcdf (kTabPolishLower, m355_elemwise.h;
  // evaluated through 1 - p they carried that same rounding as noise and their cells failed validation)
  static constexpr bool kTabPolishLower = true;
  static M355_DEV double tab_q_den(double q) { return 1.0 - q; }
  static constexpr int NP = 2;
  struct K { double alpha, beta, lga, c; };
  static M355_HD K derive(const double* p) {
    const double lga = lgamma_(p[0]);
    return K{p[0], p[1], lga, p[0] * flog(p[1]) - lga};
  }
  // k_tab: log x, scaled by a power of two near alpha / beta (x ~ beta / Gamma(alpha))
  // (shapes in [0.05, 1): the asinh-warped variable and tables of log(ppf), as for gamma -- here it is the upper tail
  // that is the power law; below 0.05 ppf(1 - 1e-9) overflows)
  static M355_HD bool tab_ok(const K& k) { return k.alpha >= 0.05 && k.alpha <= 1e4 && k.beta > 0.0 && k.beta < kInf; }
  static M355_HD bool tab_ppf_log(const K& k) { return k.alpha < 1.0; }
  // inversegamma.py:48-49 cdf = Q(alpha, beta / x), whose logarithm is logcdf (72-73); inversegamma.py:64-65 sf = exp(logsf)
  template <int FN> static constexpr TabRoute tab_route() {
    return FN == M355_CDF ? TabRoute{M355_LOGCDF, kTabEpiExp} : TabRoute{M355_LOGSF, kTabEpiExp};
  }
  static M355_DEV double tab_w(double x, const K& k) {
    return tab_warp_if(k.alpha < 1.0, flog_tab_or_nan(x * pow2_near_inv(k.beta / k.alpha)));
  }
  static M355_DEV double tab_x(double w, const K& k) {
    return fdiv(exp_(k.alpha < 1.0 ? tab_unwarp(w) : w), pow2_near_inv(k.beta / k.alpha));
  }
  // sum of logpdf with the logarithms carried as a product (m355_math.h LogProd; protocol of Dist<M355_GAMMA>::SumAcc)
  struct SumAcc {
    LogProd px;
    double lin = 0.0, exc = 0.0;
    int cnt = 0;
  };
  static M355_DEV void sum_renorm(SumAcc& s) { s.px.renorm(); }
  static M355_DEV bool sum_ok(const K& k) { return __builtin_fabs(k.alpha + 1.0) <= kSumAccMaxCoef; }
  static M355_DEV void sum_add(SumAcc& s, double x, const K& k) {
    if ((x >= kMinNormal) & (x < kInf)) {
      s.px.mul(x);
      s.lin += k.beta / x;
      ++s.cnt;
    } else {
      s.exc += logpdf(x, k);
    }
  }
  static M355_DEV double sum_finish(const SumAcc& s, const K& k) {
    return (((double)s.cnt * k.c - (k.alpha + 1.0) * s.px.log()) - s.lin) + s.exc;
  }
  static M355_DEV double logpdf(double x, const K& k) {  // inversegamma.py:76-81
    return (x <= 0.0) ? -kInf : (k.c - (k.alpha + 1.0) * flog(x)) - k.beta / x;
  }
  static M355_DEV double logsf(double x, const K& k) {  // inversegamma.py:84-85
    return (x <= 0.0) ? 0.0 : flog(gammainc_p(k.alpha, k.beta / x, k.lga));
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));                                            // inversegamma.py:56-57
    if constexpr (FN == M355_CDF) return cdf_bounds(gammainc_q(k.alpha, k.beta / x, k.lga), x, 0.0, kInf);  // inversegamma.py:48-49
    if constexpr (FN == M355_LOGCDF) return (x <= 0.0) ? -kInf : flog(gammainc_q(k.alpha, k.beta / x, k.lga));  // inversegamma.py:72-73
    if constexpr (FN == M355_LOGSF) return logsf(x, k);
    if constexpr (FN == M355_SF) return exp_(logsf(x, k));                                              // inversegamma.py:64-65
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                                                  // inversegamma.py:60-61, 52-53
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(k.beta / gammaincinv_p(k.alpha, 1.0 - q, k.lga), q, 0.0, kInf);
    }
    return kNaN;
  }
};

// ----------------------------------------------------------------------------------- chisquared
template <> struct Dist<M355_CHISQUARED> {
  static constexpr int kMinWaves = 3;  // binned kernel: 168-VGPR budget around the inlined gammaincinv_p (m355_elemwise.h)
  static constexpr int kBinned = 1;
  static constexpr int NP = 1;
  struct K { double nu, h, lgh, c; };  // h = nu/2
  static M355_HD K derive(const double* p) {
    const double h = p[0] / 2.0, lgh = lgamma_(h);
    return K{p[0], h, lgh, -lgh - (p[0] * kLog2) / 2.0};
  }
  static M355_DEV double cdf(double x, const K& k) {  // chisquared.py:40-41
    return cdf_bounds(gammainc_p(k.h, x / 2.0, k.lgh), x, 0.0, kInf);
  }
  // k_tab (m355_elemwise.h; the conditions are Dist<M355_GAMMA>'s with alpha = nu / 2)
  static M355_HD bool tab_ok(const K& k) { return k.nu >= 0.02 && k.nu <= 2e4; }
  static constexpr bool kTabOkFn = true;
  template <int FN> static M355_HD bool tab_ok_fn(const K& k) { return (FN != M355_PPF && FN != M355_ISF) || k.nu >= 0.1; }
  static M355_HD bool tab_ppf_log(const K& k) { return k.nu < 2.0; }
  // chisquared.py:40-41 cdf = P(nu/2, x/2) = exp(logcdf); chisquared.py:77-78 sf = 1 - cdf
  template <int FN> static constexpr TabRoute tab_route() {
    return FN == M355_CDF ? TabRoute{M355_LOGCDF, kTabEpiExp} : TabRoute{M355_LOGCDF, kTabEpiOneMinusExp};
  }
  static M355_DEV double tab_w(double x, const K& k) { return tab_warp_if(k.nu < 2.0, flog_tab_or_nan(x * pow2_near_inv(k.nu))); }
  static M355_DEV double tab_x(double w, const K& k) { return fdiv(exp_(k.nu < 2.0 ? tab_unwarp(w) : w), pow2_near_inv(k.nu)); }
  // sum of logpdf with the logarithms carried as a product (m355_math.h LogProd; protocol of Dist<M355_GAMMA>::SumAcc)
  struct SumAcc {
    LogProd px;
    double lin = 0.0, exc = 0.0;
    int cnt = 0;
  };
  static M355_DEV void sum_renorm(SumAcc& s) { s.px.renorm(); }
  static M355_DEV bool sum_ok(const K& k) { return __builtin_fabs(k.h - 1.0) <= kSumAccMaxCoef; }
  static M355_DEV void sum_add(SumAcc& s, double x, const K& k) {
    if ((x >= kMinNormal) & (x < kInf)) {
      s.px.mul(x);
      s.lin += x;
      ++s.cnt;
    } else {
      s.exc += eval<M355_LOGPDF>(x, k);
    }
  }
  static M355_DEV double sum_finish(const SumAcc& s, const K& k) {
    return (((k.h - 1.0) * s.px.log() - s.lin / 2.0) + (double)s.cnt * k.c) + s.exc;
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF)  // chisquared.py:69-74
      return (x < 0.0) ? -kInf : (xlogy0_l(k.h - 1.0, flog(x)) - x / 2.0) + k.c;
    if constexpr (FN == M355_PDF) return exp_((x < 0.0) ? -kInf : (xlogy0_l(k.h - 1.0, flog(x)) - x / 2.0) + k.c);  // chisquared.py:49-50
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);  // chisquared.py:77-78
    if constexpr (FN == M355_LOGCDF) {                    // chisquared.py:57-66
      if (x < 0.0) return -kInf;
      return (x < k.nu) ? flog(gammainc_p(k.h, x / 2.0, k.lgh)) : flog1p(-gammainc_q(k.h, x / 2.0, k.lgh));
    }
    if constexpr (FN == M355_LOGSF) {  // chisquared.py:85-94
      if (x < 0.0) return 0.0;
      return (x < k.nu) ? flog1p(-gammainc_p(k.h, x / 2.0, k.lgh)) : flog(gammainc_q(k.h, x / 2.0, k.lgh));
    }
    if constexpr (FN == M355_PPF || FN == M355_ISF) {  // chisquared.py:44-46, 81-82
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(2.0 * gammaincinv_p(k.h, q, k.lgh), q, 0.0, kInf);
    }
    return kNaN;
  }
};

// ---------------------------------------------------------------------- scaledinversechisquared
template <> struct Dist<M355_SCALEDINVERSECHISQUARED> {
  static constexpr int kMinWaves = 3;  // binned kernel: 168-VGPR budget around the inlined gammaincinv_p (m355_elemwise.h)
  static constexpr int kBinned = 1;
  // quantile tables: the reference takes gammaincinv at 1 - q (inversegamma.py:60-61, scaledinversechisquared.py:56-59),
  // which rounds for small q -- a staircase in q, 3e-10 of x at q = 1e-8 -- but is the smooth TRUE quantile of
  // q' = 1 - fl(1 - q).  So the lookup forms its logit from (1 - den, den) with the reference's own den (tab_q_den),
  // and the nodes below the median are true quantiles: one Newton step on logcdf (kTabPolishLower, m355_elemwise.h;
  // evaluated through 1 - p they carried that same rounding as noise and their cells failed validation)
  static constexpr bool kTabPolishLower = true;
  static M355_DEV double tab_q_den(double q) { return 1.0 - q; }
  static constexpr int NP = 2;
  struct K { double nu, tau2, h, ht, lgh, c; };  // h = nu/2, ht = h*tau2
  static M355_HD K derive(const double* p) {
    const double h = p[0] / 2.0, lgh = lgamma_(h);
    return K{p[0], p[1], h, h * p[1], lgh, (flog(p[1]) * h - lgh) + flog(h) * h};
  }
  static M355_DEV double logsf(double x, const K& k) {  // scaledinversechisquared.py:88-90
    return (x <= 0.0) ? 0.0 : flog(gammainc_p(k.h, k.ht / x, k.lgh));
  }
  // k_tab: log x, scaled by a power of two near 1 / tau2
  static M355_HD bool tab_ok(const K& k) { return k.nu >= 0.1 && k.nu <= 2e4 && k.tau2 > 0.0 && k.tau2 < kInf; }
  static M355_HD bool tab_ppf_log(const K& k) { return k.nu < 2.0; }
  // scaledinversechisquared.py:43-45 cdf = Q(h, h tau2 / x), whose logarithm is logcdf (70-72); 62-63 sf = exp(logsf)
  template <int FN> static constexpr TabRoute tab_route() {
    return FN == M355_CDF ? TabRoute{M355_LOGCDF, kTabEpiExp} : TabRoute{M355_LOGSF, kTabEpiExp};
  }
  static M355_DEV double tab_w(double x, const K& k) { return tab_warp_if(k.nu < 2.0, flog_tab_or_nan(x * pow2_near_inv(k.tau2))); }
  static M355_DEV double tab_x(double w, const K& k) { return fdiv(exp_(k.nu < 2.0 ? tab_unwarp(w) : w), pow2_near_inv(k.tau2)); }
  // sum of logpdf with the logarithms carried as a product (m355_math.h LogProd; protocol of Dist<M355_GAMMA>::SumAcc)
  struct SumAcc {
    LogProd px;
    double lin = 0.0, exc = 0.0;
    int cnt = 0;
  };
  static M355_DEV void sum_renorm(SumAcc& s) { s.px.renorm(); }
  static M355_DEV bool sum_ok(const K& k) { return __builtin_fabs(k.h + 1.0) <= kSumAccMaxCoef; }
  static M355_DEV void sum_add(SumAcc& s, double x, const K& k) {
    if ((x >= kMinNormal) & (x < kInf)) {
      s.px.mul(x);
      s.lin += k.ht / x;
      ++s.cnt;
    } else {
      s.exc += eval<M355_LOGPDF>(x, k);
    }
  }
  static M355_DEV double sum_finish(const SumAcc& s, const K& k) {
    return ((-(s.px.log() * (k.h + 1.0)) - s.lin) + (double)s.cnt * k.c) + s.exc;
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF)  // scaledinversechisquared.py:75-85
      return (x <= 0.0) ? -kInf : (-(flog(x) * (k.h + 1.0)) - k.ht / x) + k.c;
    if constexpr (FN == M355_PDF) return exp_((x <= 0.0) ? -kInf : (-(flog(x) * (k.h + 1.0)) - k.ht / x) + k.c);  // scaledinversechisquared.py:52-53
    if constexpr (FN == M355_CDF) return cdf_bounds(gammainc_q(k.h, k.ht / x, k.lgh), x, 0.0, kInf);              // scaledinversechisquared.py:43-45
    if constexpr (FN == M355_LOGCDF) return (x <= 0.0) ? -kInf : flog(gammainc_q(k.h, k.ht / x, k.lgh));          // scaledinversechisquared.py:70-72
    if constexpr (FN == M355_LOGSF) return logsf(x, k);
    if constexpr (FN == M355_SF) return exp_(logsf(x, k));  // scaledinversechisquared.py:62-63
    if constexpr (FN == M355_PPF || FN == M355_ISF) {      // scaledinversechisquared.py:56-59, 48-49
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(k.ht / gammaincinv_p(k.h, 1.0 - q, k.lgh), q, 0.0, kInf);
    }
    return kNaN;
  }
};

// --------------------------------------------------------------------------------- halfstudentt
template <> struct Dist<M355_HALFSTUDENTT> {
  static constexpr int kMinWaves = 3;  // binned kernel: 168-VGPR budget around the inlined betaincinv_ (m355_elemwise.h)
  static constexpr int kBinned = 1;
  static constexpr int NP = 2;
  struct K { double nu, sigma, logsigma, c, lbeta_h, ihalf_h; };
  static M355_HD K derive(const double* p) {
    K k;
    k.nu = p[0];
    k.sigma = p[1];
    k.logsigma = flog(p[1]);
    k.c = ((lgamma_((p[0] + 1.0) / 2.0) - lgamma_(p[0] / 2.0)) - 0.5 * flog(p[0] * kPi * (p[1] * p[1]))) + kLog2;
    k.lbeta_h = lbeta_(0.5 * p[0], 0.5);
    k.ihalf_h = betainc_half(0.5 * p[0], 0.5, k.lbeta_h);
    return k;
  }
  // ---- k_tab / k_coltab (m355_elemwise.h) ----
  // The reference's cdf is 1 - I_{x'}(nu/2, 1/2) at x' = nu / (xn^2 + nu) (halfstudentt.py:76-83), and x' rounds: next
  // to 0 the cdf is a staircase in x (1.1e-16 / (B(1/2, nu/2) sqrt(1 - x')) per step) -- but a smooth function of the
  // ROUNDED x'.  The table variable is therefore w = log(u / x') with x' formed exactly as the exact body forms it and
  // u = 1 - x' (exact for x' >= 1/2): a lookup lands on the reference's own step, and the nodes are I_u(1/2, nu/2) with
  // u supplied.  What is left is the rounding of (1 - f) 2 - 1 itself, 2e-16 absolute on the cdf: logcdf carries it
  // divided by the cdf and logsf = log1mexp(log cdf) divided by the sf, so table values below log(1e-4) resp.
  // log(4e-4) -- where that exceeds a fifth of the parity bar -- are misses (tab_value_miss: 1e-4 / 4e-4 of the mass).
  // Quantiles are judged with the modelled rounding of sqrt(nu / b - nu) (tests/harness.py _student_window): no band.
  static M355_HD bool tab_ok(const K& k) { return k.nu >= 1.0 && k.nu <= 50.0 && k.sigma > 0.0 && k.sigma < kInf; }
  static M355_DEV double tab_w(double x, const K& k) {
    const double xn = x / k.sigma;
    const double xp = k.nu / (xn * xn + k.nu);  // (IEEE divisions, as in cdf() below: this rounding IS the staircase)
    return (xn > 0.0) ? flog_tab_or_nan(fdiv_t<1, true>(1.0 - xp, xp)) : kNaN;  // (xn^2 forgets the sign: x <= 0 is a miss)
  }
  static M355_DEV double tab_x(double w, const K& k) { return k.sigma * sqrt_(k.nu * exp_(w)); }  // u / x' = xn^2 / nu
  static constexpr bool kTabNode = true;
  template <int FN> static M355_DEV double tab_node(double w, const K& k) {
    const double e = exp_(-__builtin_fabs(w));
    const double sm = fdiv(e, 1.0 + e);
    const double u = (w < 0.0) ? sm : 1.0 - sm, xp = (w < 0.0) ? 1.0 - sm : sm;
    constexpr bool UP = FN == M355_SF || FN == M355_LOGSF;
    // (<false>: nu <= 50 here, no asymptotic expansion in the table kernels' call graph -- see Dist<M355_STUDENTT>::tab_miss)
    const double v = UP ? betainc_xy<false>(0.5 * k.nu, 0.5, xp, u, k.lbeta_h) : betainc_xy<false>(0.5, 0.5 * k.nu, u, xp, k.lbeta_h);
    return (FN == M355_LOGCDF || FN == M355_LOGSF) ? flog(v) : v;
  }
  template <int FN> static M355_DEV bool tab_value_miss(double r) {
    if (FN == M355_LOGCDF) return r < -9.210340371976182;  // log(1e-4)
    if (FN == M355_LOGSF) return r < -7.824046010856292;   // log(4e-4)
    return false;
  }
  static M355_DEV bool tab_q_miss(double) { return false; }
  // halfstudentt.py:94-101 takes betaincinv at 2 - 2 qf, qf = (q + 1) / 2: q + 1 rounds (1.1e-16 absolute, i.e.
  // 1e-16 / (1 - q) of the upper-tail probability), and the reference's quantile is the smooth function of THAT
  static M355_DEV double tab_q_den(double q) { return 2.0 - 2.0 * ((q + 1.0) / 2.0); }
  // quantile at a node: p = sigmoid(-|w|) is the smaller of q, 1 - q; y = xn^2 / (xn^2 + nu) and b = 1 - y from whichever
  // of I_y(1/2, nu/2) = q, I_b(nu/2, 1/2) = 1 - q has the small right-hand side; xn = sqrt(nu y / b)
  static M355_DEV double tab_ppf_node(double w, const K& k) {
    const double e = exp_(-__builtin_fabs(w));
    const double p = fdiv(e, 1.0 + e);
    double y, b;
    if (w < 0.0) y = betaincinv_<false>(0.5, 0.5 * k.nu, p, k.lbeta_h, 1.0 - k.ihalf_h, &b);
    else b = betaincinv_<false>(0.5 * k.nu, 0.5, p, k.lbeta_h, k.ihalf_h, &y);
    return k.sigma * (sqrt_(k.nu) * sqrt_(fdiv(y, b)));
  }
  template <bool ASYM = true> static M355_DEV double cdf(double x, const K& k) {  // halfstudentt.py:76-83
    if (k.nu > 1e5) return halfnormal_cdf(x, k.sigma);
    const double xn = x / k.sigma;
    const double f = 0.5 * betainc_<ASYM>(0.5 * k.nu, 0.5, k.nu / (xn * xn + k.nu), k.lbeta_h);
    const double c = ((xn < 0.0) ? f : 1.0 - f) * 2.0 - 1.0;
    return cdf_bounds(c, x, 0.0, kInf);
  }
  static M355_DEV double logpdf(double x, const K& k) {  // halfstudentt.py:117-131
    if (k.nu > 1e5) return halfnormal_logpdf(x, k.sigma, k.logsigma);
    const double xn = x / k.sigma;
    return (x < 0.0) ? -kInf : k.c - 0.5 * (k.nu + 1.0) * flog(1.0 + (xn * xn) / k.nu) ;
  }
  template <int FN, bool ASYM = true> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));          // halfstudentt.py:90-91
    if constexpr (FN == M355_CDF) return cdf<ASYM>(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf<ASYM>(x, k);              // halfstudentt.py:104-105
    if constexpr (FN == M355_LOGCDF) return flog(cdf<ASYM>(x, k));          // halfstudentt.py:113-114
    if constexpr (FN == M355_LOGSF) return log1mexp_(flog(cdf<ASYM>(x, k)));  // halfstudentt.py:134-135
    if constexpr (FN == M355_PPF || FN == M355_ISF) {                // halfstudentt.py:94-101, 86-87
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      const double qf = (q + 1.0) / 2.0;
      double v;
      if (qf < 0.5) v = betaincinv_<ASYM>(0.5 * k.nu, 0.5, 2.0 * qf, k.lbeta_h, k.ihalf_h);
      else v = sqrt_(k.nu / betaincinv_<ASYM>(0.5 * k.nu, 0.5, 2.0 - 2.0 * qf, k.lbeta_h, k.ihalf_h) - k.nu);
      return ppf_bounds_cont(v * k.sigma, q, 0.0, kInf);
    }
    return kNaN;
  }
  template <int FN> static M355_DEV double tab_miss(double x, const K& k) { return eval<FN, false>(x, k); }  // (tab_ok: nu <= 50)
};

// ----------------------------------------------------------------------------------- betascaled
template <> struct Dist<M355_BETASCALED> {
  static constexpr int kMinWaves = 3;  // binned kernel: 168-VGPR budget around the inlined betaincinv_ (m355_elemwise.h)
  static constexpr int kBinned = 1;
  static constexpr int NP = 4;
  static constexpr bool kLogTable = true;  // logpdf uses flog_tab
  struct K { double a, b, lo, hi, w, lbeta, c, ihalf; };  // c = xlogy0(a+b-1, hi-lo) + betaln(a, b); ihalf = I_{1/2}(a, b)
  static M355_HD K derive(const double* p) {
    K k;
    k.a = p[0];
    k.b = p[1];
    k.lo = p[2];
    k.hi = p[3];
    k.w = p[3] - p[2];
    k.lbeta = lbeta_(p[0], p[1]);
    k.c = xlogy0_l(p[0] + p[1] - 1.0, flog(p[3] - p[2])) + k.lbeta;
    k.ihalf = betainc_half(p[0], p[1], k.lbeta);
    return k;
  }
  // k_tab: logit of the position in (lower, upper), both differences formed from x directly (conditions as for beta)
  static M355_HD bool tab_ok(const K& k) { return k.a >= 1.0 && k.b >= 1.0 && k.a + k.b <= 500.0 && k.w > 0.0 && k.w < kInf; }
  static M355_HD double tab_tol(const K& k) { return 1e-14 * (k.a + k.b > 40.0 ? (k.a + k.b) * (1.0 / 40.0) : 1.0); }  // (as Dist<M355_BETA>)
  // quantile cells: the root lower + p (upper - lower) is rounded on the scale of the larger end (m355_elemwise.h, tab_qfloor)
  static M355_DEV double tab_qfloor(const K& k) {
    const double al = __builtin_fabs(k.lo), ah = __builtin_fabs(k.hi);
    return al > ah ? al : ah;
  }
  // betascaled.py:71-72 cdf = exp(logcdf); betascaled.py:88-89 sf = 1 - cdf
  template <int FN> static constexpr TabRoute tab_route() {
    return FN == M355_CDF ? TabRoute{M355_LOGCDF, kTabEpiExp} : TabRoute{M355_LOGCDF, kTabEpiOneMinusExp};
  }
  static M355_DEV double tab_w(double x, const K& k) { return flog_tab_or_nan(fdiv_t<1, true>(x - k.lo, k.hi - x)); }
  static M355_DEV double tab_x(double w, const K& k) { return __builtin_fma(k.w, fdiv(1.0, 1.0 + exp_(-w)), k.lo); }
  // Node values from the position p = sigmoid(w) in (0, 1) and its complement, both formed directly (as Dist<M355_BETA>):
  // x = lower + p (upper - lower) rounds at 1.1e-16 max(|lower|, |upper|), which next to an end of the support is a
  // relative error of the position itself -- 1.5e-12 at p = 2e-5 on (-1, 3), i.e. 3e-12 in logcdf with alpha = 2 (round 3:
  // tests/test_gpu_tables_full.py::test_tabulated_cdf_sf_relative_accuracy measured 1.6e-12) -- while the reference's
  // (x - lower) / (upper - lower) of a DATA point is exact to an ulp of the position.
  static constexpr bool kTabNode = true;
  template <int FN> static M355_DEV double tab_node(double w, const K& k) {
    const double e = exp_(-__builtin_fabs(w));
    const double sm = fdiv(e, 1.0 + e);
    const double p = (w < 0.0) ? sm : 1.0 - sm, q = (w < 0.0) ? 1.0 - sm : sm;
    constexpr bool UP = FN == M355_LOGSF || FN == M355_SF;
    const double I = UP ? betainc_xy<false>(k.b, k.a, q, p, k.lbeta) : betainc_xy<false>(k.a, k.b, p, q, k.lbeta);  // (as Dist<M355_BETA>)
    if (FN == M355_LOGCDF || FN == M355_LOGSF) return flog(I);
    return (FN == M355_SF) ? 1.0 - betainc_xy<false>(k.a, k.b, p, q, k.lbeta) : I;  // betascaled.py:88-89: sf = 1 - cdf
  }
  // sum of logpdf: sum log(x - lo), sum log(hi - x) as LogProds (protocol of Dist<M355_GAMMA>::SumAcc)
  struct SumAcc {
    LogProd px, pc;
    double exc = 0.0;
    int cnt = 0;
  };
  static M355_DEV void sum_renorm(SumAcc& s) {
    s.px.renorm();
    s.pc.renorm();
  }
  static M355_DEV bool sum_ok(const K& k) {
    return (__builtin_fabs(k.a - 1.0) <= kSumAccMaxCoef) & (__builtin_fabs(k.b - 1.0) <= kSumAccMaxCoef);
  }
  static M355_DEV void sum_add(SumAcc& s, double x, const K& k) {
    const double l = x - k.lo, r = k.hi - x;
    if ((l >= kMinNormal) & (r >= kMinNormal) & (l < kInf) & (r < kInf)) {
      s.px.mul(l);
      s.pc.mul(r);
      ++s.cnt;
    } else {
      s.exc += logpdf(x, k);
    }
  }
  static M355_DEV double sum_finish(const SumAcc& s, const K& k) {
    return (((k.a - 1.0) * s.px.log() + (k.b - 1.0) * s.pc.log()) - (double)s.cnt * k.c) + s.exc;
  }
  static M355_DEV double logpdf(double x, const K& k) {  // betascaled.py:106-112
    if ((x < k.lo) | (x > k.hi)) return -kInf;
    return (xlogy0_l(k.a - 1.0, flog_tab(x - k.lo)) + xlogy0_l(k.b - 1.0, flog_tab(k.hi - x))) - k.c;
  }
  template <bool ASYM = true> static M355_DEV double logcdf(double x, const K& k) {  // betascaled.py:97-103
    if (x < k.lo) return -kInf;
    if (x > k.hi) return 0.0;
    return flog(betainc_<ASYM>(k.a, k.b, (x - k.lo) / k.w, k.lbeta));
  }
  template <int FN, bool ASYM = true> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));        // betascaled.py:79-80
    if constexpr (FN == M355_LOGCDF) return logcdf<ASYM>(x, k);
    if constexpr (FN == M355_CDF) return exp_(logcdf<ASYM>(x, k));        // betascaled.py:71-72
    if constexpr (FN == M355_SF) return 1.0 - exp_(logcdf<ASYM>(x, k));   // betascaled.py:88-89
    if constexpr (FN == M355_LOGSF) {                               // betascaled.py:115-121
      if (x < k.lo) return 0.0;
      if (x > k.hi) return -kInf;
      return flog(betainc_<ASYM>(k.b, k.a, 1.0 - (x - k.lo) / k.w, k.lbeta));
    }
    if constexpr (FN == M355_PPF || FN == M355_ISF) {  // betascaled.py:83-85, 75-76
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(betaincinv_<ASYM>(k.a, k.b, q, k.lbeta, k.ihalf) * k.w + k.lo, q, k.lo, k.hi);
    }
    return kNaN;
  }
  template <int FN> static M355_DEV double tab_miss(double x, const K& k) { return eval<FN, false>(x, k); }  // (as Dist<M355_BETA>)
};

}  // namespace m355
