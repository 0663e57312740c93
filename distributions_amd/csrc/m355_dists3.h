// Scalar bodies of the discrete families: bernoulli, geometric,
// discreteuniform, discreteweibull, negativebinomial, betabinomial, hypergeometric and the
// zero-inflated poisson / binomial / negativebinomial.  Same Dist<D>::derive / eval<FN> template;
// observations arrive as doubles (int64 data is converted exactly below 2^53), every support test
// is an exact comparison.  Quantiles that the reference finds with find_ppf_discrete
// (optimization.py:102-129) run that search per element (m355_dists.h).  CDFs that the reference
// writes as explicit sums over arange (betabinomial.py, hypergeometric.py) are per-element loops in
// the same ascending order.
#pragma once
#include "m355_dists2.h"

namespace m355 {

// ------------------------------------------------------------------------------------ bernoulli
template <> struct Dist<M355_BERNOULLI> {
  static constexpr int NP = 1;
  struct K { double p, logp, log1mp; };
  static M355_HD K derive(const double* p) { return K{p[0], flog(p[0]), flog(1.0 - p[0])}; }
  static M355_DEV double cdf(double x, const K& k) { return cdf_bounds((x < 1.0) ? 1.0 - k.p : 1.0, x, 0.0, 1.0); }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_PDF) return (x == 1.0) ? k.p : ((x == 0.0) ? 1.0 - k.p : 0.0);
    if constexpr (FN == M355_LOGPDF) return (x == 1.0) ? k.logp : ((x == 0.0) ? k.log1mp : -kInf);
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_LOGCDF) return (x < 0.0) ? -kInf : ((x < 1.0) ? k.log1mp : 0.0);
    if constexpr (FN == M355_SF) return (x < 0.0) ? 1.0 : ((x < 1.0) ? k.p : 0.0);
    if constexpr (FN == M355_LOGSF) return flog1p(-cdf(x, k));
    if constexpr (FN == M355_PPF || FN == M355_ISF) {
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_disc((q <= 1.0 - k.p) ? 0.0 : 1.0, q, 0.0, 1.0);
    }
    return kNaN;
  }
};

// ------------------------------------------------------------------------------------ geometric
template <> struct Dist<M355_GEOMETRIC> {
  static constexpr int NP = 1;
  struct K { double p, logp, log1mp; };
  static M355_HD K derive(const double* p) { return K{p[0], flog(p[0]), flog1p(-p[0])}; }
  static M355_DEV double cdf(double x, const K& k) {
    return cdf_bounds(-expm1_(k.log1mp * floor_(x)), x, 1.0, kInf);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return (x < 1.0) ? -kInf : (x - 1.0) * k.log1mp + k.logp;
    if constexpr (FN == M355_PDF) return exp_((x < 1.0) ? -kInf : (x - 1.0) * k.log1mp + k.logp);
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);
    if constexpr (FN == M355_LOGCDF) return flog(cdf(x, k));
    if constexpr (FN == M355_LOGSF) return flog1p(-cdf(x, k));
    if constexpr (FN == M355_PPF || FN == M355_ISF) {
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      const double vals = __builtin_ceil(flog1p(-q) / k.log1mp);
      const double temp = cdf(vals - 1.0, k);
      const double r = ((temp >= q) & (vals > 0.0)) ? vals - 1.0 : vals;
      return ppf_bounds_disc(r, q, 1.0, kInf);
    }
    return kNaN;
  }
};

// ------------------------------------------------------------------------------ discreteuniform
template <> struct Dist<M355_DISCRETEUNIFORM> {
  static constexpr int NP = 2;
  struct K { double lo, hi, n, mlogn; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], p[1] - p[0] + 1.0, -flog(p[1] - p[0] + 1.0)}; }
  static M355_DEV double cdf(double x, const K& k) {
    return cdf_bounds((floor_(x) - k.lo + 1.0) / (k.hi - k.lo + 1.0), x, k.lo, k.hi);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return ((x >= k.lo) & (x <= k.hi)) ? k.mlogn : -kInf;
    if constexpr (FN == M355_PDF) return exp_(((x >= k.lo) & (x <= k.hi)) ? k.mlogn : -kInf);
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);
    if constexpr (FN == M355_LOGCDF) return flog(cdf(x, k));
    if constexpr (FN == M355_LOGSF) return flog1p(-cdf(x, k));
    if constexpr (FN == M355_PPF || FN == M355_ISF) {
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_disc(k.lo + __builtin_ceil(q * k.n) - 1.0, q, k.lo, k.hi);
    }
    return kNaN;
  }
};

// ------------------------------------------------------------------------------ discreteweibull
template <> struct Dist<M355_DISCRETEWEIBULL> {
  static constexpr int NP = 2;
  struct K { double q, beta, logq; };
  static M355_HD K derive(const double* p) { return K{p[0], p[1], flog(p[0])}; }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    const double xf = floor_(x);
    if constexpr (FN == M355_CDF) return cdf_bounds(1.0 - pow_(k.q, pow_(x + 1.0, k.beta)), x, 0.0, kInf);
    if constexpr (FN == M355_PDF)
      return (x < 0.0) ? 0.0 : pow_(k.q, pow_(xf, k.beta)) - pow_(k.q, pow_(xf + 1.0, k.beta));
    if constexpr (FN == M355_SF) return (x < 0.0) ? 1.0 : pow_(k.q, pow_(xf + 1.0, k.beta));
    if constexpr (FN == M355_LOGCDF) return (x < 0.0) ? -kInf : flog(1.0 - pow_(k.q, pow_(xf + 1.0, k.beta)));
    if constexpr (FN == M355_LOGPDF)
      return (x < 0.0) ? -kInf : flog(pow_(k.q, pow_(xf, k.beta)) - pow_(k.q, pow_(xf + 1.0, k.beta)));
    if constexpr (FN == M355_LOGSF) return (x < 0.0) ? 0.0 : flog(pow_(k.q, pow_(xf + 1.0, k.beta)));
    if constexpr (FN == M355_PPF || FN == M355_ISF) {
      const double p = (FN == M355_ISF) ? 1.0 - x : x;
      const double v = __builtin_ceil(pow_(flog(1.0 - p) / k.logq, 1.0 / k.beta) - 1.0);
      return ppf_bounds_disc(v, p, 0.0, kInf);
    }
    return kNaN;
  }
};

// ----------------------------------------------------------------------------- negativebinomial
struct NegBinK {
  double n, p, lgn, c, log1mp;  // c = xlogy0(n, p) - gammaln(n)
};
M355_HD NegBinK negbin_derive(double n, double p) {
  const double lgn = lgamma_(n);
  return NegBinK{n, p, lgn, xlogy0_l(n, flog(p)) - lgn, flog(1.0 - p)};
}
M355_DEV double negbin_logpdf(double x, const NegBinK& k) {
  if (x < 0.0) return -kInf;
  return ((lgamma_obs(x + k.n) - lgamma1p_obs(x)) + k.c) + xlogy0_l(x, k.log1mp);
}
M355_DEV double negbin_cdf(double x, const NegBinK& k) {
  const double kk = floor_(x);
  if (kk < 0.0) return 0.0;
  if (kk >= kInf) return 1.0;
  if (!(kk == kk)) return kNaN;
  return betainc_(k.n, kk + 1.0, k.p, lbeta_(k.n, kk + 1.0));
}
M355_DEV double negbin_sf(double x, const NegBinK& k) {
  const double kk = floor_(x);
  if (kk < 0.0) return 1.0;
  if (kk > kInf) return 0.0;
  if (kk == kInf) return 0.0;
  if (!(kk == kk)) return kNaN;
  return betainc_(kk + 1.0, k.n, 1.0 - k.p, lbeta_(kk + 1.0, k.n));
}
template <> struct Dist<M355_NEGATIVEBINOMIAL> {
  static constexpr int kBinned = 1;
  static constexpr int NP = 2;
  using K = NegBinK;
  static M355_HD bool dtab_range(const K& k, double& lo, int& len) {
    return dtab_window(k.n * (1.0 - k.p) / k.p, sqrt(k.n * (1.0 - k.p)) / k.p, 0.0, kInf, lo, len);
  }
  static M355_HD K derive(const double* p) { return negbin_derive(p[0], p[1]); }
  // the pieces of the quantile search, named so that k_dtab can run the same search on tabulated values
  static M355_DEV double dtab_cdf(double v, const K& k) { return negbin_cdf(v, k); }
  static M355_DEV double dtab_pmf(double v, const K& k) { return exp_(negbin_logpdf(v, k)); }
  template <class C, class P> static M355_DEV double ppf_with(double q, const K& k, C c, P p) {
    return find_ppf_discrete(q, k.n * (1.0 - k.p) / k.p, 0.0, kInf, c, p);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return negbin_logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(negbin_logpdf(x, k));
    if constexpr (FN == M355_CDF) return negbin_cdf(x, k);
    if constexpr (FN == M355_SF) return negbin_sf(x, k);
    if constexpr (FN == M355_LOGCDF) return flog(negbin_cdf(x, k));
    if constexpr (FN == M355_LOGSF) return flog(negbin_sf(x, k));
    if constexpr (FN == M355_PPF || FN == M355_ISF) {
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_with(q, k, [&](double v) { return dtab_cdf(v, k); }, [&](double v) { return dtab_pmf(v, k); });
    }
    return kNaN;
  }
};

// --------------------------------------------------------------------------------- betabinomial
template <> struct Dist<M355_BETABINOMIAL> {
  static constexpr int kBinned = 1;
  static constexpr int NP = 3;
  struct K { double n, a, b, c; };  // c: the x-independent gammaln terms
  static M355_HD bool dtab_range(const K& k, double& lo, int& len) {  // the whole support when it fits, else around the mean
    const double m = k.n * k.a / (k.a + k.b);
    const double v = m * k.b / (k.a + k.b) * (k.a + k.b + k.n) / (k.a + k.b + 1.0);
    return dtab_window(m, k.n < (double)kDTabMax ? k.n : sqrt(v), 0.0, k.n, lo, len);
  }
  static M355_HD K derive(const double* p) {
    const double n = p[0], a = p[1], b = p[2];
    return K{n, a, b, (((lgamma_(n + 1.0) - lgamma_(n + a + b)) + lgamma_(a + b)) - lgamma_(a)) - lgamma_(b)};
  }
  static M355_DEV double logpdf(double x, const K& k) {
    if ((x < 0.0) | (x > k.n)) return -kInf;
    return (((k.c - lgamma1p_obs(x)) - lgamma_obs(k.n - x + 1.0)) + lgamma_obs(x + k.a)) + lgamma_obs(k.n - x + k.b);
  }
  static M355_DEV double cdf(double x, const K& k) {
    if (x < 0.0) return 0.0;
    if (x >= k.n) return 1.0;
    if (!(x == x)) return kNaN;
    const double top = floor_(x);
    double s = 0.0;
    for (double j = 0.0; j <= top; j += 1.0) s += exp_(logpdf(j, k));
    return s;
  }
  // the pieces of the quantile search, named so that k_dtab can run the same search on tabulated values
  static M355_DEV double dtab_cdf(double v, const K& k) { return cdf(v, k); }
  static M355_DEV double dtab_pmf(double v, const K& k) { return exp_(logpdf(v, k)); }
  template <class C, class P> static M355_DEV double ppf_with(double q, const K& k, C c, P p) {
    return find_ppf_discrete(q, k.n * k.a / (k.a + k.b), 0.0, k.n, c, p);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);
    if constexpr (FN == M355_LOGCDF) return flog(cdf(x, k));
    if constexpr (FN == M355_LOGSF) return flog(1.0 - cdf(x, k));
    if constexpr (FN == M355_PPF || FN == M355_ISF) {
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_with(q, k, [&](double v) { return dtab_cdf(v, k); }, [&](double v) { return dtab_pmf(v, k); });
    }
    return kNaN;
  }
};

// ------------------------------------------------------------------------------- hypergeometric
template <> struct Dist<M355_HYPERGEOMETRIC> {
  static constexpr int kBinned = 1;
  static constexpr int NP = 3;
  struct K { double N, k, n, lo, hi, lbNn, lgk1, lgNk1; };
  static M355_HD bool dtab_range(const K& k, double& lo, int& len) {
    const double m = k.n * k.k / k.N;
    return dtab_window(m, sqrt(m * (1.0 - k.k / k.N)), k.lo, k.hi, lo, len);
  }
  static M355_DEV double logbinom(double n, double k) { return (lgamma_(n + 1.0) - lgamma_(k + 1.0)) - lgamma_(n - k + 1.0); }
  static M355_HD K derive(const double* p) {
    const double N = p[0], k = p[1], n = p[2];
    const double lo = (n + k - N > 0.0) ? n + k - N : 0.0, hi = k < n ? k : n;
    return K{N, k, n, lo, hi, (lgamma_(N + 1.0) - lgamma_(n + 1.0)) - lgamma_(N - n + 1.0), lgamma_(k + 1.0),
             lgamma_(N - k + 1.0)};
  }
  static M355_DEV double logpdf(double x, const K& k) {
    const double xf = floor_(x);
    if (!((xf >= k.lo) & (xf <= k.hi))) return -kInf;
    const double a = (k.lgk1 - lgamma1p_obs(xf)) - lgamma_obs(k.k - xf + 1.0);
    const double nx = k.n - xf;
    const double b = (k.lgNk1 - lgamma1p_obs(nx)) - lgamma_obs((k.N - k.k) - nx + 1.0);
    return (a + b) - k.lbNn;
  }
  static M355_DEV double cdf(double x, const K& k) {
    if (!(x == x)) return kNaN;
    if (x < k.lo) return 0.0;
    if (x >= k.hi) return 1.0;
    const double top = floor_(x);
    double s = 0.0;
    for (double j = 0.0; j <= top; j += 1.0) s += exp_(logpdf(j, k));
    return s;
  }
  // the pieces of the quantile search, named so that k_dtab can run the same search on tabulated values
  static M355_DEV double dtab_cdf(double v, const K& k) { return cdf(v, k); }
  static M355_DEV double dtab_pmf(double v, const K& k) { return exp_(logpdf(v, k)); }
  template <class C, class P> static M355_DEV double ppf_with(double q, const K& k, C c, P p) {
    return find_ppf_discrete(q, k.n * k.k / k.N, k.lo, k.hi, c, p);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);
    if constexpr (FN == M355_LOGCDF) return flog(cdf(x, k));
    if constexpr (FN == M355_LOGSF) return flog1p(-cdf(x, k));
    if constexpr (FN == M355_PPF || FN == M355_ISF) {
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_with(q, k, [&](double v) { return dtab_cdf(v, k); }, [&](double v) { return dtab_pmf(v, k); });
    }
    return kNaN;
  }
};

// ------------------------------------------------------------------------------- zero-inflated
// Shared shape of zi_poisson.py / zi_binomial.py / zi_negativebinomial.py: BASE supplies the base
// distribution's logpdf / cdf / sf and the zero probability.
template <> struct Dist<M355_ZI_POISSON> {
  static constexpr int kBinned = 1;
  static constexpr int NP = 2;
  struct K { double psi, logpsi, log_zero; Dist<M355_POISSON>::K b; };
  static M355_HD bool dtab_range(const K& k, double& lo, int& len) {  // the base distribution's window, down to 0
    return dtab_window(0.5 * k.b.mu, 0.5 * k.b.mu / 9.0 + sqrt(k.b.mu), 0.0, kInf, lo, len);
  }
  static M355_HD K derive(const double* p) {
    return K{p[0], flog(p[0]), flog((1.0 - p[0]) + p[0] * ::exp(-p[1])), Dist<M355_POISSON>::derive(p + 1)};
  }
  static M355_DEV double logpdf(double x, const K& k) {
    if (x == kInf) return kNaN;
    if ((x < 0.0) | (x == -kInf)) return -kInf;
    return (x == 0.0) ? k.log_zero : k.logpsi + Dist<M355_POISSON>::logpdf(x, k.b);
  }
  static M355_DEV double pdf(double x, const K& k) { return (x == kInf) ? kNaN : exp_(logpdf(x, k)); }
  static M355_DEV double cdf(double x, const K& k) {
    return cdf_bounds((1.0 - k.psi) + k.psi * Dist<M355_POISSON>::cdf(x, k.b), x, 0.0, kInf);
  }
  static M355_DEV double sf(double x, const K& k) { return sf_bounds(k.psi * Dist<M355_POISSON>::sf(x, k.b), x, 0.0, kInf); }
  // the pieces of the quantile search, named so that k_dtab can run the same search on tabulated values
  static M355_DEV double dtab_cdf(double v, const K& k) { return cdf(v, k); }
  static M355_DEV double dtab_pmf(double v, const K& k) { return pdf(v, k); }
  template <class C, class P> static M355_DEV double ppf_with(double q, const K& k, C c, P p) {
    return find_ppf_discrete(q, k.b.mu, 0.0, kInf, c, p);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return pdf(x, k);
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return sf(x, k);
    if constexpr (FN == M355_LOGCDF) {
      if (x < 0.0) return -kInf;
      if (x == kInf) return 0.0;
      return flog1p(k.psi * (Dist<M355_POISSON>::cdf(x, k.b) - 1.0));
    }
    if constexpr (FN == M355_LOGSF) return flog(sf(x, k));
    if constexpr (FN == M355_PPF || FN == M355_ISF) {
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_with(q, k, [&](double v) { return dtab_cdf(v, k); }, [&](double v) { return dtab_pmf(v, k); });
    }
    return kNaN;
  }
};

template <> struct Dist<M355_ZI_BINOMIAL> {
  static constexpr int kBinned = 1;
  static constexpr int NP = 3;
  struct K { double psi, logpsi, log_zero; Dist<M355_BINOMIAL>::K b; };
  static M355_HD bool dtab_range(const K& k, double& lo, int& len) {
    const double m = k.b.n * k.b.p;
    return dtab_window(0.5 * m, 0.5 * m / 9.0 + sqrt(m * (1.0 - k.b.p)), 0.0, k.b.n, lo, len);
  }
  static M355_HD K derive(const double* p) {
    return K{p[0], flog(p[0]), flog((1.0 - p[0]) + p[0] * ::pow(1.0 - p[2], p[1])), Dist<M355_BINOMIAL>::derive(p + 1)};
  }
  static M355_DEV double logpdf(double x, const K& k) {
    if ((x < 0.0) | (x > k.b.n)) return -kInf;
    return (x == 0.0) ? k.log_zero : k.logpsi + Dist<M355_BINOMIAL>::logpdf(x, k.b);
  }
  static M355_DEV double cdf(double x, const K& k) {
    return cdf_bounds((1.0 - k.psi) + k.psi * Dist<M355_BINOMIAL>::cdf(x, k.b), x, 0.0, k.b.n);
  }
  // the pieces of the quantile search, named so that k_dtab can run the same search on tabulated values
  static M355_DEV double dtab_cdf(double v, const K& k) { return cdf(v, k); }
  static M355_DEV double dtab_pmf(double v, const K& k) { return exp_(logpdf(v, k)); }
  template <class C, class P> static M355_DEV double ppf_with(double q, const K& k, C c, P p) {
    return find_ppf_discrete(q, k.psi * k.b.n * k.b.p, 0.0, k.b.n, c, p);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);
    if constexpr (FN == M355_LOGCDF) {
      if (x < 0.0) return -kInf;
      if (x > k.b.n) return 0.0;
      return flog1p(k.psi * (Dist<M355_BINOMIAL>::cdf(x, k.b) - 1.0));
    }
    if constexpr (FN == M355_LOGSF) return flog1p(-cdf(x, k));
    if constexpr (FN == M355_PPF || FN == M355_ISF) {
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_with(q, k, [&](double v) { return dtab_cdf(v, k); }, [&](double v) { return dtab_pmf(v, k); });
    }
    return kNaN;
  }
};

template <> struct Dist<M355_ZI_NEGATIVEBINOMIAL> {
  static constexpr int kBinned = 1;
  static constexpr int NP = 3;
  struct K { double psi, logpsi, log_zero; NegBinK b; };
  static M355_HD bool dtab_range(const K& k, double& lo, int& len) {
    const double m = k.b.n * (1.0 - k.b.p) / k.b.p;
    return dtab_window(0.5 * m, 0.5 * m / 9.0 + sqrt(k.b.n * (1.0 - k.b.p)) / k.b.p, 0.0, kInf, lo, len);
  }
  static M355_HD K derive(const double* p) {
    return K{p[0], flog(p[0]), flog((1.0 - p[0]) + p[0] * ::pow(p[2], p[1])), negbin_derive(p[1], p[2])};
  }
  static M355_DEV double logpdf(double x, const K& k) {
    if (x == kInf) return kNaN;
    if ((x < 0.0) | (x == -kInf)) return -kInf;
    return (x == 0.0) ? k.log_zero : k.logpsi + negbin_logpdf(x, k.b);
  }
  static M355_DEV double pdf(double x, const K& k) { return (x == kInf) ? kNaN : exp_(logpdf(x, k)); }
  static M355_DEV double cdf(double x, const K& k) {
    return cdf_bounds((1.0 - k.psi) + k.psi * negbin_cdf(x, k.b), x, 0.0, kInf);
  }
  static M355_DEV double sf(double x, const K& k) { return sf_bounds(k.psi * negbin_sf(x, k.b), x, 0.0, kInf); }
  // the pieces of the quantile search, named so that k_dtab can run the same search on tabulated values
  static M355_DEV double dtab_cdf(double v, const K& k) { return cdf(v, k); }
  static M355_DEV double dtab_pmf(double v, const K& k) { return pdf(v, k); }
  template <class C, class P> static M355_DEV double ppf_with(double q, const K& k, C c, P p) {
    return find_ppf_discrete(q, k.psi * k.b.n * (1.0 - k.b.p) / k.b.p, 0.0, kInf, c, p);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return pdf(x, k);
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return sf(x, k);
    if constexpr (FN == M355_LOGCDF) {
      if (x < 0.0) return -kInf;
      if (x == kInf) return 0.0;
      return flog1p(k.psi * (negbin_cdf(x, k.b) - 1.0));
    }
    if constexpr (FN == M355_LOGSF) return flog(sf(x, k));
    if constexpr (FN == M355_PPF || FN == M355_ISF) {
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_with(q, k, [&](double v) { return dtab_cdf(v, k); }, [&](double v) { return dtab_pmf(v, k); });
    }
    return kNaN;
  }
};

}  // namespace m355
