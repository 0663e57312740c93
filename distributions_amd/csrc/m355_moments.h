// Moment functions of every module of the reference: mean, mode, median, var, std, skewness, kurtosis, entropy
// (SURVEY.md 8(f) rank 4 lists the quadrature helpers behind them, helper.py:99-322).  One value per parameter
// set; Mom<D>::eval<C>(stat, p) follows the reference's formula of that module (cited per specialisation).
//
// C is the summation context of the quadrature / explicit-sum helpers: SerialCtx (one thread per parameter
// set: closed forms, and the host build for CPU unit tests) or WaveCtx (one 64-lane wavefront per parameter set:
// the 1000-point trapezoids of continuous_moment / continuous_entropy and the sums over the support of
// discrete_entropy / discrete_moment are spread over the lanes and combined by a butterfly of DPP/shuffle adds,
// so every lane ends with the same total and the closed-form part runs uniformly).
#pragma once
#include "../../include/m355_dist.h"
#include "m355_dists4.h"
#include "m355_grad.h"  // digamma_
#include "m355_hermgauss70.h"

namespace m355 {

#if defined(__HIPCC__)
M355_DEV double tgamma_(double x) { return ::tgamma(x); }
M355_DEV double sin_(double x) { return ::sin(x); }
M355_DEV double cos_(double x) { return ::cos(x); }
M355_DEV double tanh_(double x) { return ::tanh(x); }
M355_DEV double cosh_(double x) { return ::cosh(x); }
#else
M355_DEV double tgamma_(double x) { return std::tgamma(x); }
M355_DEV double sin_(double x) { return std::sin(x); }
M355_DEV double cos_(double x) { return std::cos(x); }
M355_DEV double tanh_(double x) { return std::tanh(x); }
M355_DEV double cosh_(double x) { return std::cosh(x); }
#endif

struct SerialCtx {
  // sum_{i = 0}^{n - 1} f(i)
  template <class F> static M355_DEV double sum(long long n, F f) {
    double s = 0.0;
    for (long long i = 0; i < n; ++i) s += f(i);
    return s;
  }
  // index of the first maximum of f(0..n-1) (numpy.argmax: a NaN counts as the maximum)
  template <class F> static M355_DEV long long argmax(long long n, F f) {
    long long bi = 0;
    double bv = f(0);
    for (long long i = 1; i < n; ++i) {
      const double v = f(i);
      if ((bv == bv) && (!(v == v) || v > bv)) {
        bv = v;
        bi = i;
      }
    }
    return bi;
  }
};
#if defined(__HIPCC__)
struct WaveCtx {
  template <class F> static M355_DEV double sum(long long n, F f) {
    double s = 0.0;
    for (long long i = threadIdx.x & 63; i < n; i += 64) s += f(i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
  }
  template <class F> static M355_DEV long long argmax(long long n, F f) {
    const long long lane = threadIdx.x & 63;
    long long bi = (lane < n) ? lane : 0;
    double bv = f(bi);
    for (long long i = lane + 64; i < n; i += 64) {
      const double v = f(i);
      if ((bv == bv) && (!(v == v) || v > bv)) {
        bv = v;
        bi = i;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(bv, o, 64);
      const long long oi = __shfl_xor(bi, o, 64);
      const bool on = !(ov == ov), mn = !(bv == bv);
      // the other side wins when it is "larger" (NaN largest), or equal with the smaller index
      const bool gt = mn ? false : (on || ov > bv);
      const bool eq = (on && mn) || (ov == bv);
      if (gt || (eq && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    return bi;
  }
};
#endif

// <dist>.<fn>(x, *p) of the elementwise path, for the medians and the quadrature integrands
template <int D, int FN> M355_DEV double dist_fn(double x, const double* p) {
  return Dist<D>::template eval<FN>(x, Dist<D>::derive(p));
}

// ------------------------------------------------------------------ helper.py:99-322, one parameter set
// trapezoid over linspace(lo, hi, n): dx * (f_0 / 2 + sum f_1..f_{n-2} + f_{n-1} / 2), x_i = lo + i * dx
template <class C, class F> M355_DEV double trapezoid(double lo, double hi, int n, F f) {
  const double dx = (hi - lo) / (double)(n - 1);
  const double inner = C::sum(n - 2, [&](long long i) { return f(lo + (double)(i + 1) * dx); });
  return dx * ((0.5 * f(lo) + inner) + 0.5 * f(hi));
}
constexpr int kQuadPoints = 1000;
// helper.py:99-142
template <int D, class C> M355_DEV double continuous_entropy(double lo, double hi, const double* p) {
  const auto k = Dist<D>::derive(p);
  return trapezoid<C>(lo, hi, kQuadPoints, [&](double x) {
    const double lp = Dist<D>::template eval<M355_LOGPDF>(x, k);
    return -exp_(lp) * lp;
  });
}
// helper.py:242-295: raw moment (central == false) or central moment about `mean`
template <int D, class C> M355_DEV double continuous_moment(double lo, double hi, const double* p, int order, bool central, double mean) {
  const auto k = Dist<D>::derive(p);
  return trapezoid<C>(lo, hi, kQuadPoints, [&](double x) {
    const double pdf = exp_(Dist<D>::template eval<M355_LOGPDF>(x, k));
    const double b = central ? x - mean : x;
    double w = b;
    for (int j = 1; j < order; ++j) w *= b;
    return w * pdf;
  });
}
template <int D, class C> M355_DEV double continuous_skewness(double lo, double hi, const double* p) {  // helper.py:309-314
  const double m = continuous_moment<D, C>(lo, hi, p, 1, false, 0.0);
  const double v = continuous_moment<D, C>(lo, hi, p, 2, true, m);
  const double t = continuous_moment<D, C>(lo, hi, p, 3, true, m);
  const double s = sqrt_(v);
  return t / (s * s * s);
}
template <int D, class C> M355_DEV double continuous_kurtosis(double lo, double hi, const double* p) {  // helper.py:317-322
  const double m = continuous_moment<D, C>(lo, hi, p, 1, false, 0.0);
  const double v = continuous_moment<D, C>(lo, hi, p, 2, true, m);
  const double t = continuous_moment<D, C>(lo, hi, p, 4, true, m);
  return t / (v * v) - 3.0;
}
// helper.py:145-176: sum over k in arange(lo, hi) of -exp(lp) lp
template <int D, class C> M355_DEV double discrete_entropy(double lo, double hi, const double* p) {
  const auto k = Dist<D>::derive(p);
  const long long n = (long long)__builtin_ceil(hi - lo);
  if (!(n > 0)) return 0.0;
  return C::sum(n, [&](long long i) {
    const double lp = Dist<D>::template eval<M355_LOGPDF>(lo + (double)i, k);
    return -exp_(lp) * lp;
  });
}
// helper.py:179-214: the range is [ppf(0.0001), ppf(0.9999)] of the parameter set
template <int D, class C> M355_DEV double discrete_moment(const double* p, int order) {
  const auto k = Dist<D>::derive(p);
  const double lo = Dist<D>::template eval<M355_PPF>(0.0001, k);
  const double hi = Dist<D>::template eval<M355_PPF>(0.9999, k) + 1.0;
  const long long n = (long long)__builtin_ceil(hi - lo);
  if (!(n > 0)) return 0.0;
  const double mean = C::sum(n, [&](long long i) {
    const double x = lo + (double)i;
    return x * Dist<D>::template eval<M355_PDF>(x, k);
  });
  if (order == 1) return mean;
  return C::sum(n, [&](long long i) {
    const double x = lo + (double)i;
    const double b = x - mean;
    double w = b;
    for (int j = 1; j < order; ++j) w *= b;
    return w * Dist<D>::template eval<M355_PDF>(x, k);
  });
}

template <int D> struct Mom;  // eval<C>(stat, p); kWave: stats (bit mask) that sum over many points

#define M355_MOM_HEAD(mask) static constexpr unsigned kWave = (mask); template <class C> static M355_DEV double eval(int s, const double* p)
constexpr unsigned mbit(int s) { return 1u << s; }
constexpr double kLog2PiE = 2.8378770664093454836;  // log(2 pi e)

template <> struct Mom<M355_NORMAL> {  // normal.py:6-43
  M355_MOM_HEAD(0) {
    const double mu = p[0], sg = p[1];
    switch (s) {
      case M355_MEAN: case M355_MODE: case M355_MEDIAN: return mu;
      case M355_VAR: return sg * sg;
      case M355_STD: return sg;
      case M355_SKEWNESS: case M355_KURTOSIS: return 0.0;
      case M355_ENTROPY: return 0.5 * flog(2.0 * kPi * 2.718281828459045 * (sg * sg));
    }
    return kNaN;
  }
};

template <> struct Mom<M355_GAMMA> {  // gamma.py:8-37
  M355_MOM_HEAD(0) {
    const double a = p[0], b = p[1];
    switch (s) {
      case M355_MEAN: return a / b;
      case M355_MODE: return (a <= 1.0) ? 0.0 : (a - 1.0) / b;
      case M355_MEDIAN: return dist_fn<M355_GAMMA, M355_PPF>(0.5, p);
      case M355_VAR: return a / (b * b);
      case M355_STD: return sqrt_(a) / b;
      case M355_SKEWNESS: return 2.0 / sqrt_(a);
      case M355_KURTOSIS: return 6.0 / a;
      case M355_ENTROPY: return ((a - flog(b)) + lgamma_(a)) + (1.0 - a) * digamma_(a);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_BETA> {  // beta.py:9-63
  M355_MOM_HEAD(0) {
    const double a = p[0], b = p[1];
    const double psc = a + b, prod = a * b;
    switch (s) {
      case M355_MEAN: return a / (a + b);
      case M355_MODE: {
        double r = kNaN;
        if ((a == 1.0) & (b > 1.0)) r = 0.0;
        if ((b == 1.0) & (a > 1.0)) r = 1.0;
        if ((a > 1.0) & (b > 1.0)) r = (a - 1.0) / (a + b - 2.0);
        return r;
      }
      case M355_MEDIAN: return dist_fn<M355_BETA, M355_PPF>(0.5, p);
      case M355_VAR: return prod / ((psc * psc) * (psc + 1.0));
      case M355_STD: return sqrt_(prod / ((psc * psc) * (psc + 1.0)));
      case M355_SKEWNESS: return (a == b) ? 0.0 : (2.0 * (b - a) * sqrt_(psc + 1.0)) / ((psc + 2.0) * sqrt_(prod));
      case M355_KURTOSIS: {
        const double d = __builtin_fabs(a - b);
        return 6.0 * (d * d * (psc + 1.0) - prod * (psc + 2.0)) / (prod * (psc + 2.0) * (psc + 3.0));
      }
      case M355_ENTROPY:
        return ((lbeta_(a, b) - (a - 1.0) * digamma_(a)) - (b - 1.0) * digamma_(b)) + (psc - 2.0) * digamma_(psc);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_STUDENTT> {  // studentt.py:11-64
  static M355_DEV double var(double nu, double sg) {
    double r = kNaN;
    if (nu > 2.0) r = sg * sg * nu / (nu - 2.0);
    if ((nu > 1.0) & (nu <= 2.0)) r = kInf;
    if (nu > 1e10) r = sg * sg;
    return r;
  }
  M355_MOM_HEAD(0) {
    const double nu = p[0], mu = p[1], sg = p[2];
    switch (s) {
      case M355_MEAN: return (nu > 1.0) ? mu : kNaN;
      case M355_MODE: case M355_MEDIAN: return mu;
      case M355_VAR: return var(nu, sg);
      case M355_STD: {
        const double v = var(nu, sg);
        return (v - v == 0.0) ? sqrt_(v) : v;
      }
      case M355_SKEWNESS: return (nu > 3.0) ? 0.0 : kNaN;
      case M355_KURTOSIS: {
        double r = kNaN;
        if (nu > 4.0) r = 6.0 / (nu - 4.0);
        if ((nu > 2.0) & (nu <= 4.0)) r = kInf;
        return r;
      }
      case M355_ENTROPY:
        if (nu > 1e5) return 0.5 * flog(2.0 * kPi * 2.718281828459045 * (sg * sg));
        return ((flog(sg) + 0.5 * (nu + 1.0) * (digamma_(0.5 * (nu + 1.0)) - digamma_(0.5 * nu))) + flog(sqrt_(nu))) +
               lbeta_(0.5 * nu, 0.5);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_POISSON> {  // poisson.py:9-43
  M355_MOM_HEAD(mbit(M355_ENTROPY)) {
    const double mu = p[0];
    switch (s) {
      case M355_MEAN: case M355_VAR: return mu;
      case M355_MODE: case M355_MEDIAN: return floor_(mu);
      case M355_STD: return sqrt_(mu);
      case M355_SKEWNESS: return 1.0 / sqrt_(mu);
      case M355_KURTOSIS: return 1.0 / mu;
      case M355_ENTROPY:
        if (mu < 10.0) return discrete_entropy<M355_POISSON, C>(0.0, 25.0, p);
        return 0.5 * flog(2.0 * kPi * 2.718281828459045 * mu) - 1.0 / (12.0 * mu);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_BINOMIAL> {  // binomial.py:9-38
  M355_MOM_HEAD(mbit(M355_ENTROPY)) {
    const double n = p[0], q = p[1];
    const double v = n * q * (1.0 - q);
    switch (s) {
      case M355_MEAN: return n * q;
      case M355_MODE: return floor_((n + 1.0) * q);
      case M355_MEDIAN: return dist_fn<M355_BINOMIAL, M355_PPF>(0.5, p);
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: return ((1.0 - q) - q) / sqrt_(v);
      case M355_KURTOSIS: return (1.0 - 6.0 * q * (1.0 - q)) / v;
      case M355_ENTROPY: return discrete_entropy<M355_BINOMIAL, C>(0.0, n + 1.0, p);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_EXPONENTIAL> {  // exponential.py:5-38
  M355_MOM_HEAD(0) {
    const double lam = p[0];
    switch (s) {
      case M355_MEAN: case M355_STD: return 1.0 / lam;
      case M355_MODE: return 0.0;
      case M355_MEDIAN: return 0.69314718055994530942 / lam;
      case M355_VAR: return 1.0 / (lam * lam);
      case M355_SKEWNESS: return 2.0;
      case M355_KURTOSIS: return 6.0;
      case M355_ENTROPY: return 1.0 - flog(lam);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_LOGNORMAL> {  // lognormal.py:6-42
  M355_MOM_HEAD(0) {
    const double mu = p[0], sg = p[1];
    const double s2 = sg * sg;
    switch (s) {
      case M355_MEAN: return exp_(mu + s2 / 2.0);
      case M355_MODE: return exp_(mu - s2);
      case M355_MEDIAN: return exp_(mu);
      case M355_VAR: return (exp_(s2) - 1.0) * exp_(2.0 * mu + s2);
      case M355_STD: return sqrt_((exp_(s2) - 1.0) * exp_(2.0 * mu + s2));
      case M355_SKEWNESS: return (exp_(s2) + 2.0) * sqrt_(exp_(s2) - 1.0);
      case M355_KURTOSIS: return ((exp_(4.0 * s2) + 2.0 * exp_(3.0 * s2)) + 3.0 * exp_(2.0 * s2)) - 6.0;
      case M355_ENTROPY: return flog(2.5066282746310002 * sg * exp_(mu + 0.5));
    }
    return kNaN;
  }
};

template <> struct Mom<M355_HALFNORMAL> {  // halfnormal.py:5-35
  M355_MOM_HEAD(0) {
    const double sg = p[0];
    switch (s) {
      case M355_MEAN: return sg * 0.79788456080286535588;  // sqrt(2 / pi)
      case M355_MODE: return 0.0;
      case M355_MEDIAN: return sg * 1.4142135623730951 * 0.4769362762044699;  // sqrt(2) erfinv(0.5)
      case M355_VAR: return sg * sg * (1.0 - 2.0 / kPi);
      case M355_STD: return sg * sqrt_(1.0 - 2.0 / kPi);
      case M355_SKEWNESS: return 0.9952717;
      case M355_KURTOSIS: return 0.8691773;
      case M355_ENTROPY: return 0.5 * flog(kPi * (sg * sg) / 2.0) + 0.5;
    }
    return kNaN;
  }
};

template <> struct Mom<M355_CAUCHY> {  // cauchy.py:6-41
  M355_MOM_HEAD(0) {
    switch (s) {
      case M355_MODE: case M355_MEDIAN: return p[0];
      case M355_ENTROPY: return flog(4.0 * kPi * p[1]);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_HALFCAUCHY> {  // halfcauchy.py:6-35
  M355_MOM_HEAD(0) {
    switch (s) {
      case M355_MEAN: case M355_VAR: case M355_STD: return kInf;
      case M355_MODE: return 0.0;
      case M355_MEDIAN: return dist_fn<M355_HALFCAUCHY, M355_PPF>(0.5, p);
      case M355_ENTROPY: return flog(2.0 * kPi * p[0]);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_LAPLACE> {  // laplace.py:6-43
  M355_MOM_HEAD(0) {
    const double mu = p[0], b = p[1];
    switch (s) {
      case M355_MEAN: case M355_MODE: case M355_MEDIAN: return mu;
      case M355_VAR: return 2.0 * (b * b);
      case M355_STD: return 1.4142135623730951 * b;
      case M355_SKEWNESS: return 0.0;
      case M355_KURTOSIS: return 3.0;
      case M355_ENTROPY: return flog(2.0 * b) + 1.0;
    }
    return kNaN;
  }
};

template <> struct Mom<M355_LOGISTIC> {  // logistic.py:6-43
  M355_MOM_HEAD(0) {
    const double mu = p[0], sc = p[1];
    switch (s) {
      case M355_MEAN: case M355_MODE: case M355_MEDIAN: return mu;
      case M355_VAR: return sc * sc * (kPi * kPi) / 3.0;
      case M355_STD: return sc * kPi / 1.7320508075688772;
      case M355_SKEWNESS: return 0.0;
      case M355_KURTOSIS: return 1.2;
      case M355_ENTROPY: return flog(sc) + 2.0;
    }
    return kNaN;
  }
};

template <> struct Mom<M355_GUMBEL> {  // gumbel.py:6-41
  M355_MOM_HEAD(0) {
    const double mu = p[0], b = p[1];
    switch (s) {
      case M355_MEAN: return mu + b * kEuler;
      case M355_MODE: return mu;
      case M355_MEDIAN: return mu - b * -0.3665129;
      case M355_VAR: return 1.64493407 * (b * b);
      case M355_STD: return b * 1.2825498;
      case M355_SKEWNESS: return 1.1395471;
      case M355_KURTOSIS: return 2.4;
      case M355_ENTROPY: return (flog(b) + 1.0) + kEuler;
    }
    return kNaN;
  }
};

template <> struct Mom<M355_WEIBULL> {  // weibull.py:8-44
  static M355_DEV double mean(double a, double b) { return b * tgamma_(1.0 + 1.0 / a); }
  static M355_DEV double var(double a, double b) {
    const double m = mean(a, b);
    return b * b * tgamma_(1.0 + 2.0 / a) - m * m;
  }
  static M355_DEV double skew(double a, double b) {
    const double sg = sqrt_(var(a, b));
    const double m_s = mean(a, b) / sg, r = b / sg;
    return (tgamma_(1.0 + 3.0 / a) * (r * r * r) - 3.0 * m_s) - m_s * m_s * m_s;
  }
  M355_MOM_HEAD(0) {
    const double a = p[0], b = p[1];
    switch (s) {
      case M355_MEAN: return mean(a, b);
      case M355_MODE: return (a > 1.0) ? b * pow_((a - 1.0) / a, 1.0 / a) : 0.0;
      case M355_MEDIAN: return b * pow_(0.69314718055994530942, 1.0 / a);
      case M355_VAR: return var(a, b);
      case M355_STD: return sqrt_(var(a, b));
      case M355_SKEWNESS: return skew(a, b);
      case M355_KURTOSIS: {
        const double sg = sqrt_(var(a, b));
        const double m_s = mean(a, b) / sg, r = b / sg;
        const double m2 = m_s * m_s;
        return ((((r * r) * (r * r) * tgamma_(1.0 + 4.0 / a) - 4.0 * skew(a, b) * m_s) - 6.0 * m2) - m2 * m2) - 3.0;
      }
      case M355_ENTROPY: return (kEuler * (1.0 - 1.0 / a) + flog(b / a)) + 1.0;
    }
    return kNaN;
  }
};

template <> struct Mom<M355_PARETO> {  // pareto.py:5-44
  M355_MOM_HEAD(0) {
    const double a = p[0], m = p[1];
    switch (s) {
      case M355_MEAN: return (a > 1.0) ? a * m / (a - 1.0) : kInf;
      case M355_MODE: return m;
      case M355_MEDIAN: return m * pow_(2.0, 1.0 / a);
      case M355_VAR: return (a > 2.0) ? m * m * a / (((a - 1.0) * (a - 1.0)) * (a - 2.0)) : kInf;
      case M355_STD: return sqrt_((a > 2.0) ? m * m * a / (((a - 1.0) * (a - 1.0)) * (a - 2.0)) : kInf);
      case M355_SKEWNESS: return (a > 3.0) ? 2.0 * (1.0 + a) / (a - 3.0) * sqrt_(1.0 - 2.0 / a) : kNaN;
      case M355_KURTOSIS:
        return (a > 4.0) ? 6.0 * (((a * a * a + a * a) - 6.0 * a) - 2.0) / (a * (a - 3.0) * (a - 4.0)) : kNaN;
      case M355_ENTROPY: return flog(m / a) + (1.0 + 1.0 / a);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_UNIFORM> {  // uniform.py:5-38
  M355_MOM_HEAD(0) {
    const double lo = p[0], up = p[1];
    switch (s) {
      case M355_MEAN: case M355_MEDIAN: return (up + lo) / 2.0;
      case M355_MODE: return kNaN;
      case M355_VAR: return (up - lo) * (up - lo) / 12.0;
      case M355_STD: return sqrt_((up - lo) * (up - lo) / 12.0);
      case M355_SKEWNESS: return 0.0;
      case M355_KURTOSIS: return -6.0 / 5.0;
      case M355_ENTROPY: return flog(up - lo);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_TRIANGULAR> {  // triangular.py:5-41
  M355_MOM_HEAD(0) {
    const double lo = p[0], c = p[1], up = p[2];
    const double q = ((((lo * lo + up * up) + c * c) - lo * c) - c * up) - lo * up;
    switch (s) {
      case M355_MEAN: return (lo + c + up) / 3.0;
      case M355_MODE: return c;
      case M355_MEDIAN:
        return (c >= (lo + up) / 2.0) ? lo + sqrt_((up - lo) * (c - lo) / 2.0) : up - sqrt_((up - lo) * (up - c) / 2.0);
      case M355_VAR: return q / 18.0;
      case M355_STD: return sqrt_(q / 18.0);
      case M355_SKEWNESS:
        return (1.4142135623730951 * (lo + up - 2.0 * c) * (2.0 * lo - up - c) * (lo - 2.0 * up + c)) / (5.0 * (q * sqrt_(q)));
      case M355_KURTOSIS: return -3.0 / 5.0;
      case M355_ENTROPY: return 0.5 + flog((up - lo) / 2.0);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_KUMARASWAMY> {  // kumaraswamy.py:6-50
  static M355_DEV double raw(double k, double a, double b) {
    return b * exp_((lgamma_(1.0 + k / a) + lgamma_(b)) - lgamma_(1.0 + k / a + b));
  }
  M355_MOM_HEAD(0) {
    const double a = p[0], b = p[1];
    const double m1 = raw(1.0, a, b);
    switch (s) {
      case M355_MEAN: return m1;
      case M355_MODE: return ((a > 1.0) | (b > 1.0)) ? pow_((a - 1.0) / (a * b - 1.0), 1.0 / a) : 0.5;
      case M355_MEDIAN: return dist_fn<M355_KUMARASWAMY, M355_PPF>(0.5, p);
    }
    const double m2 = raw(2.0, a, b);
    const double v = m2 - m1 * m1;
    switch (s) {
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: return ((raw(3.0, a, b) - 3.0 * m1 * v) - m1 * m1 * m1) / (v * sqrt_(v));
      case M355_KURTOSIS:
        return (raw(4.0, a, b) + m1 * (-4.0 * raw(3.0, a, b) + m1 * (6.0 * m2 - 3.0 * (m1 * m1)))) / (v * v) - 3.0;
      case M355_ENTROPY: return (((1.0 - 1.0 / b) + (1.0 - 1.0 / a) * (digamma_(b + 1.0) + kEuler)) - flog(a)) - flog(b);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_MOYAL> {  // moyal.py:5-40
  M355_MOM_HEAD(0) {
    const double mu = p[0], sg = p[1];
    switch (s) {
      case M355_MEAN: return mu + sg * (kEuler + 0.69314718055994530942);
      case M355_MODE: return mu;
      case M355_MEDIAN: return mu + sg * 0.7875976;
      case M355_VAR: return (sg * sg) * (kPi * kPi) / 2.0;
      case M355_STD: return sqrt_((sg * sg) * (kPi * kPi) / 2.0);
      case M355_SKEWNESS: return 1.5351416;
      case M355_KURTOSIS: return 4.0;
      case M355_ENTROPY: return flog(sg) + 2.0541199;
    }
    return kNaN;
  }
};

template <> struct Mom<M355_ASYMMETRICLAPLACE> {  // asymmetriclaplace.py:5-40
  M355_MOM_HEAD(0) {
    const double mu = p[0], b = p[1], k = p[2];
    const double k2 = k * k, k4 = k2 * k2;
    const double v = ((1.0 / k) * (1.0 / k) + k2) * (b * b);
    switch (s) {
      case M355_MEAN: return (1.0 / k - k) * b + mu;
      case M355_MODE: return mu;
      case M355_MEDIAN:
        return (k > 1.0) ? mu + k * b * flog((1.0 + k2) / (2.0 * k2)) : mu - flog((1.0 + k2) / 2.0) / (k / b);
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: return 2.0 * (1.0 - k4 * k2) / ((1.0 + k4) * sqrt_(1.0 + k4));
      case M355_KURTOSIS: return 6.0 * (1.0 + k4 * k4) / ((1.0 + k4) * (1.0 + k4));
      case M355_ENTROPY: return (1.0 + flog(k + 1.0 / k)) + flog(b);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_LOGLOGISTIC> {  // loglogistic.py:5-63
  M355_MOM_HEAD(0) {
    const double a = p[0], be = p[1];
    const double b = kPi / be;
    const double m1 = b / sin_(b), m2 = 2.0 * b / sin_(2.0 * b);
    const double mu = a * m1, sigma2 = a * a * (m2 - m1 * m1);
    const double sg = sqrt_(sigma2);
    switch (s) {
      case M355_MEAN: return (be > 1.0) ? a * b / sin_(b) : kNaN;
      case M355_MODE: return (be > 1.0) ? a * pow_((be - 1.0) / (be + 1.0), 1.0 / be) : 0.0;
      case M355_MEDIAN: return a;
      case M355_VAR: return (be > 2.0) ? a * a * (2.0 * b / sin_(2.0 * b) - b * b / (sin_(b) * sin_(b))) : kNaN;
      case M355_STD: return sqrt_((be > 2.0) ? a * a * (2.0 * b / sin_(2.0 * b) - b * b / (sin_(b) * sin_(b))) : kNaN);
      case M355_ENTROPY: return (flog(a) - flog(be)) + 2.0;
    }
    const double m3 = 3.0 * b / sin_(3.0 * b);
    const double skew = ((a * a * a * m3 - 3.0 * mu * sigma2) - mu * mu * mu) / (sg * sg * sg);
    if (s == M355_SKEWNESS) return (be > 3.0) ? skew : kNaN;
    if (s == M355_KURTOSIS) {
      const double m4 = 4.0 * b / sin_(4.0 * b);
      const double m_s = mu / sg, r = a / sg, ms2 = m_s * m_s;
      const double res = ((((r * r) * (r * r) * m4 - 4.0 * skew * m_s) - 6.0 * ms2) - ms2 * ms2) - 3.0;
      return (be > 4.0) ? res : kNaN;
    }
    return kNaN;
  }
};


template <> struct Mom<M355_LOGITNORMAL> {  // logitnormal.py:12-92: 70-point Gauss-Hermite sums
  // sum_i w_i g(sigmoid(sqrt(2) sigma x_i + mu)) / sqrt(pi)
  template <class C, class G> static M355_DEV double ghq(double mu, double sg, G g) {
    constexpr double kX[70] = {M355_GH70_NODES};
    constexpr double kW[70] = {M355_GH70_WEIGHTS};
    return C::sum(70, [&](long long i) { return kW[i] * g(sigmoid_(1.4142135623730951 * sg * kX[i] + mu)); }) / 1.7724538509055159;
  }
  template <class C> static M355_DEV double central(double mu, double sg, int order, double mean) {
    return ghq<C>(mu, sg, [&](double x) {
      const double b = x - mean;
      double w = b;
      for (int j = 1; j < order; ++j) w *= b;
      return w;
    });
  }
  M355_MOM_HEAD(mbit(M355_MEAN) | mbit(M355_VAR) | mbit(M355_STD) | mbit(M355_SKEWNESS) | mbit(M355_KURTOSIS) | mbit(M355_ENTROPY)) {
    const double mu = p[0], sg = p[1];
    switch (s) {
      case M355_MODE: case M355_MEDIAN: return sigmoid_(mu);
      case M355_ENTROPY: {
        const auto k = Dist<M355_LOGITNORMAL>::derive(p);
        return ghq<C>(mu, sg, [&](double x) { return -Dist<M355_LOGITNORMAL>::template eval<M355_LOGPDF>(x, k); });
      }
    }
    const double m = ghq<C>(mu, sg, [](double x) { return x; });
    if (s == M355_MEAN) return m;
    const double v = central<C>(mu, sg, 2, m);
    switch (s) {
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: {
        const double sd = sqrt_(v);
        return central<C>(mu, sg, 3, m) / (sd * sd * sd);
      }
      case M355_KURTOSIS: return central<C>(mu, sg, 4, m) / (v * v) - 3.0;
    }
    return kNaN;
  }
};

template <> struct Mom<M355_INVERSEGAMMA> {  // inversegamma.py:7-45
  M355_MOM_HEAD(0) {
    const double a = p[0], b = p[1];
    const double v = (a > 2.0) ? b * b / (((a - 1.0) * (a - 1.0)) * (a - 2.0)) : kInf;
    switch (s) {
      case M355_MEAN: return (a > 1.0) ? b / (a - 1.0) : kInf;
      case M355_MODE: return b / (a + 1.0);
      case M355_MEDIAN: return dist_fn<M355_INVERSEGAMMA, M355_PPF>(0.5, p);
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: return (a > 3.0) ? 4.0 * sqrt_(a - 2.0) / (a - 3.0) : kNaN;
      case M355_KURTOSIS: return (a > 4.0) ? 6.0 * (5.0 * a - 11.0) / ((a - 3.0) * (a - 4.0)) : kNaN;
      case M355_ENTROPY: {
        if (a >= 200.0) {
          const double i1 = 1.0 / a, i2 = i1 * i1;
          return ((((((1.0 - 3.0 * flog(a)) + 0.69314718055994530942) + 1.1447298858494002) / 2.0 + (2.0 / 3.0) * i1) + i2 / 12.0) -
                  i2 * i1 / 90.0) - i2 * i2 / 120.0 + flog(b);
        }
        return ((a - (a + 1.0) * digamma_(a)) + lgamma_(a)) + flog(b);
      }
    }
    return kNaN;
  }
};

template <> struct Mom<M355_CHISQUARED> {  // chisquared.py:6-37
  M355_MOM_HEAD(0) {
    const double nu = p[0], h = p[0] / 2.0;
    switch (s) {
      case M355_MEAN: return nu;
      case M355_MODE: return (nu - 2.0 > 0.0) ? nu - 2.0 : ((nu == nu) ? 0.0 : kNaN);
      case M355_MEDIAN: return dist_fn<M355_CHISQUARED, M355_PPF>(0.5, p);
      case M355_VAR: return nu * 2.0;
      case M355_STD: return sqrt_(nu * 2.0);
      case M355_SKEWNESS: return sqrt_(8.0 / nu);
      case M355_KURTOSIS: return 12.0 / nu;
      case M355_ENTROPY: return ((h + 0.69314718055994530942) + lgamma_(h)) + (1.0 - h) * digamma_(h);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_SCALEDINVERSECHISQUARED> {  // scaledinversechisquared.py:6-40
  M355_MOM_HEAD(0) {
    const double nu = p[0], t2 = p[1], h = p[0] / 2.0;
    const double v = (nu > 4.0) ? 2.0 * (nu * nu) * (t2 * t2) / (((nu - 2.0) * (nu - 2.0)) * (nu - 4.0)) : kInf;
    switch (s) {
      case M355_MEAN: return (nu > 2.0) ? nu * t2 / (nu - 2.0) : kInf;
      case M355_MODE: return nu * t2 / (nu + 2.0);
      case M355_MEDIAN: return dist_fn<M355_SCALEDINVERSECHISQUARED, M355_PPF>(0.5, p);
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: return (nu > 6.0) ? 4.0 / (nu - 6.0) * sqrt_(2.0 * (nu - 4.0)) : kNaN;
      case M355_KURTOSIS: return (nu > 8.0) ? (12.0 * (5.0 * nu - 22.0)) / ((nu - 6.0) * (nu - 8.0)) : kNaN;
      case M355_ENTROPY: return ((h + flog(h * t2)) + lgamma_(h)) - (1.0 + h) * digamma_(h);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_HALFSTUDENTT> {  // halfstudentt.py:11-73
  static M355_DEV double var(double nu, double sg) {
    const double g0 = exp_(lgamma_((nu + 1.0) / 2.0)), g1 = exp_(lgamma_(nu / 2.0));
    const bool bad = (g0 == kInf) | (g1 == kInf);
    const double r = g0 / g1;
    const double fin = sg * sg * ((nu / (nu - 2.0)) - ((4.0 * nu) / (kPi * ((nu - 1.0) * (nu - 1.0)))) * (r * r));
    return (nu > 2.0) ? (bad ? sg * sg * (1.0 - 2.0 / kPi) : fin) : kInf;
  }
  M355_MOM_HEAD(0) {
    const double nu = p[0], sg = p[1];
    switch (s) {
      case M355_MEAN: {
        const double g0 = exp_(lgamma_((nu + 1.0) / 2.0)), g1 = exp_(lgamma_(nu / 2.0));
        const bool bad = (g0 == kInf) | (g1 == kInf);
        const double fin = 2.0 * sg * sqrt_(nu / kPi) * (g0 / (g1 * (nu - 1.0)));
        return (nu > 1.0) ? (bad ? sg * 0.79788456080286535588 : fin) : kInf;
      }
      case M355_MODE: return 0.0;
      case M355_MEDIAN: return dist_fn<M355_HALFSTUDENTT, M355_PPF>(0.5, p);
      case M355_VAR: return var(nu, sg);
      case M355_STD: {
        const double v = var(nu, sg);
        return (v - v == 0.0) ? sqrt_(v) : v;
      }
      case M355_ENTROPY:
        if (nu > 1e5) return 0.5 * flog(kPi * (sg * sg) / 2.0) + 0.5;
        return (((flog(sg) + 0.5 * (nu + 1.0) * (digamma_(0.5 * (nu + 1.0)) - digamma_(0.5 * nu))) + flog(sqrt_(nu))) +
                lbeta_(0.5 * nu, 0.5)) - 0.69314718055994530942;
    }
    return kNaN;
  }
};

template <> struct Mom<M355_BETASCALED> {  // betascaled.py:9-68
  M355_MOM_HEAD(0) {
    const double a = p[0], b = p[1], lo = p[2], up = p[3];
    const double psc = a + b, prod = a * b;
    const double v = prod / ((psc * psc) * (psc + 1.0)) * ((up - lo) * (up - lo));
    switch (s) {
      case M355_MEAN: return (a * up + b * lo) / (a + b);
      case M355_MODE: {
        double r = kNaN;
        if ((a <= 1.0) & (b > 1.0)) r = lo;
        if ((b <= 1.0) & (a > 1.0)) r = up;
        if ((a > 1.0) & (b > 1.0)) r = lo + (a - 1.0) / (a + b - 2.0) * (up - lo);
        return r;
      }
      case M355_MEDIAN: return dist_fn<M355_BETASCALED, M355_PPF>(0.5, p);
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: case M355_KURTOSIS: return Mom<M355_BETA>::eval<C>(s, p);
      case M355_ENTROPY: return Mom<M355_BETA>::eval<C>(s, p) + flog(up - lo);
    }
    return kNaN;
  }
};


M355_DEV double xlogx_(double x) { return (x == 0.0) ? 0.0 : x * flog(x); }  // pytensor.tensor.xlogx.xlogx

template <> struct Mom<M355_BERNOULLI> {  // bernoulli.py:8-38
  M355_MOM_HEAD(0) {
    const double q = p[0], v = p[0] * (1.0 - p[0]);
    switch (s) {
      case M355_MEAN: return q;
      case M355_MODE: case M355_MEDIAN: return (q <= 0.5) ? 0.0 : 1.0;
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: return ((1.0 - q) - q) / sqrt_(v);
      case M355_KURTOSIS: return (1.0 - 6.0 * q * (1.0 - q)) / v;
      case M355_ENTROPY: return -xlogx_(q) - xlogx_(1.0 - q);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_GEOMETRIC> {  // geometric.py:6-35
  M355_MOM_HEAD(0) {
    const double q = p[0];
    switch (s) {
      case M355_MEAN: return 1.0 / q;
      case M355_MODE: return 1.0;
      case M355_MEDIAN: return dist_fn<M355_GEOMETRIC, M355_PPF>(0.5, p);
      case M355_VAR: return (1.0 - q) / (q * q);
      case M355_STD: return sqrt_((1.0 - q) / (q * q));
      case M355_SKEWNESS: return (2.0 - q) / sqrt_(1.0 - q);
      case M355_KURTOSIS: return 6.0 + (q * q) / (1.0 - q);
      case M355_ENTROPY: return (-(1.0 - q) * flog(1.0 - q) - q * flog(q)) / q;
    }
    return kNaN;
  }
};

template <> struct Mom<M355_DISCRETEUNIFORM> {  // discreteuniform.py:6-38
  M355_MOM_HEAD(mbit(M355_ENTROPY)) {
    const double lo = p[0], up = p[1];
    const double n = (up - lo) + 1.0;
    switch (s) {
      case M355_MEAN: return (up + lo) / 2.0;
      case M355_MODE: return kNaN;
      case M355_MEDIAN: return floor_((up + lo) / 2.0);
      case M355_VAR: return (n * n - 1.0) / 12.0;
      case M355_STD: return sqrt_((n * n - 1.0) / 12.0);
      case M355_SKEWNESS: return 0.0;
      case M355_KURTOSIS: return -(6.0 * (n * n + 1.0)) / (5.0 * (n * n - 1.0));
      case M355_ENTROPY: return discrete_entropy<M355_DISCRETEUNIFORM, C>(lo, up + 1.0, p);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_DISCRETEWEIBULL> {  // discreteweibull.py:14-46
  M355_MOM_HEAD(mbit(M355_MEAN) | mbit(M355_VAR) | mbit(M355_STD) | mbit(M355_SKEWNESS) | mbit(M355_KURTOSIS) | mbit(M355_ENTROPY)) {
    constexpr int D = M355_DISCRETEWEIBULL;
    switch (s) {
      case M355_MEAN: return discrete_moment<D, C>(p, 1);
      case M355_MODE: return floor_(-flog(-flog(p[0])) / flog(p[1]));
      case M355_MEDIAN: return dist_fn<D, M355_PPF>(0.5, p);
      case M355_VAR: return discrete_moment<D, C>(p, 2);
      case M355_STD: return sqrt_(discrete_moment<D, C>(p, 2));
      case M355_SKEWNESS: {  // helper.py:227-231
        const double v = discrete_moment<D, C>(p, 2);
        return discrete_moment<D, C>(p, 3) / (v * sqrt_(v));
      }
      case M355_KURTOSIS: {  // helper.py:234-239
        const double v = discrete_moment<D, C>(p, 2);
        return discrete_moment<D, C>(p, 4) / (v * v) - 3.0;
      }
      case M355_ENTROPY: {
        const double q = dist_fn<D, M355_PPF>(0.9999, p);
        return discrete_entropy<D, C>(0.0, (q < 1e4) ? q : ((q == q) ? 1e4 : q), p);
      }
    }
    return kNaN;
  }
};

template <> struct Mom<M355_NEGATIVEBINOMIAL> {  // negativebinomial.py:7-39
  M355_MOM_HEAD(mbit(M355_ENTROPY)) {
    const double n = p[0], q = p[1];
    switch (s) {
      case M355_MEAN: return n * (1.0 - q) / q;
      case M355_MODE: return (n < 1.0) ? 0.0 : floor_((n - 1.0) * (1.0 - q) / q);
      case M355_MEDIAN: return dist_fn<M355_NEGATIVEBINOMIAL, M355_PPF>(0.5, p);
      case M355_VAR: return n * (1.0 - q) / (q * q);
      case M355_STD: return sqrt_(n * (1.0 - q) / (q * q));
      case M355_SKEWNESS: return (2.0 - q) / sqrt_(n * (1.0 - q));
      case M355_KURTOSIS: return 6.0 / n + (q * q) / (n * (1.0 - q));
      case M355_ENTROPY:
        return discrete_entropy<M355_NEGATIVEBINOMIAL, C>(dist_fn<M355_NEGATIVEBINOMIAL, M355_PPF>(0.0001, p),
                                                          dist_fn<M355_NEGATIVEBINOMIAL, M355_PPF>(0.9999, p), p);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_BETABINOMIAL> {  // betabinomial.py:10-80
  M355_MOM_HEAD(mbit(M355_ENTROPY)) {
    const double n = p[0], a = p[1], b = p[2];
    const double sm = a + b, pr = a * b;
    const double v = n * a * b * (sm + n) / ((sm * sm) * (sm + 1.0));
    switch (s) {
      case M355_MEAN: return n * a / sm;
      case M355_MODE: {
        double r = kNaN;
        if ((a == 1.0) & (b > 1.0)) r = 0.0;
        if ((a > 1.0) & (b == 1.0)) r = n;
        if ((a > 1.0) & (b > 1.0)) {
          const double m = floor_((n + 1.0) * ((a - 1.0) / (sm - 2.0)));
          r = (m < 0.0) ? 0.0 : ((m > n) ? n : m);
        }
        return r;
      }
      case M355_MEDIAN: return dist_fn<M355_BETABINOMIAL, M355_PPF>(0.5, p);
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: return (sm + 2.0 * n) * (b - a) / (sm + 2.0) * sqrt_((1.0 + a + b) / (n * a * b * (sm + n)));
      case M355_KURTOSIS: {
        const double left = ((sm * sm) * (1.0 + sm)) / ((n * pr) * (sm + 2.0) * (sm + 3.0) * (sm + n));
        double right = (sm * ((sm - 1.0) + 6.0 * n) + 3.0 * pr * (n - 2.0)) + 6.0 * (n * n);
        right -= (3.0 * pr * n * (6.0 - n)) / sm;
        right -= (18.0 * pr * (n * n)) / (sm * sm);
        return left * right - 3.0;
      }
      case M355_ENTROPY: return discrete_entropy<M355_BETABINOMIAL, C>(0.0, n + 1.0, p);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_HYPERGEOMETRIC> {  // hypergeometric.py:8-60
  M355_MOM_HEAD(mbit(M355_ENTROPY)) {
    const double N = p[0], k = p[1], n = p[2];
    const double v = n * k * (N - k) * (N - n) / (N * N * (N - 1.0));
    switch (s) {
      case M355_MEAN: return n * k / N;
      case M355_MODE: return floor_((n + 1.0) * (k + 1.0) / (N + 2.0));
      case M355_MEDIAN: return dist_fn<M355_HYPERGEOMETRIC, M355_PPF>(0.5, p);
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: return ((N - 2.0 * k) * sqrt_(N - 1.0) * (N - 2.0 * n)) / (sqrt_(n * k * (N - k) * (N - n)) * (N - 2.0));
      case M355_KURTOSIS: {
        const double m = N - k;
        const double num = N * N * (N - 1.0) * ((N * (N + 1.0) - 6.0 * n * (N - n)) - 6.0 * k * m) + 6.0 * k * n * (N - n) * m * (5.0 * N - 6.0);
        return num / (k * n * (N - n) * m * (N - 2.0) * (N - 3.0));
      }
      case M355_ENTROPY: {
        const double lo = (n + k - N > 0.0) ? n + k - N : 0.0, hi = ((k < n) ? k : n) + 1.0;
        return discrete_entropy<M355_HYPERGEOMETRIC, C>(lo, hi, p);
      }
    }
    return kNaN;
  }
};


// helper.py:425-446: 0 or the base distribution's mode, whichever has the larger probability
template <int D> M355_DEV double zi_mode(double base_mode, const double* p) {
  return (dist_fn<D, M355_LOGPDF>(0.0, p) >= dist_fn<D, M355_LOGPDF>(base_mode, p)) ? 0.0 : base_mode;
}
// skewness (ex4 < 0 ... not used) / excess kurtosis from raw moments, as zi_*.py spell them
M355_DEV double skew_from_raw(double ex1, double ex2, double ex3) {
  const double mu2 = ex2 - ex1 * ex1;
  const double mu3 = (ex3 - 3.0 * ex1 * ex2) + 2.0 * (ex1 * ex1 * ex1);
  return mu3 / (mu2 * sqrt_(mu2));
}
M355_DEV double kurt_from_raw(double ex1, double ex2, double ex3, double ex4) {
  const double mu2 = ex2 - ex1 * ex1;
  const double mu4 = ((ex4 - 4.0 * ex1 * ex3) + 6.0 * (ex1 * ex1) * ex2) - 3.0 * ((ex1 * ex1) * (ex1 * ex1));
  return mu4 / (mu2 * mu2) - 3.0;
}

template <> struct Mom<M355_ZI_POISSON> {  // zi_poisson.py:9-59
  M355_MOM_HEAD(mbit(M355_ENTROPY)) {
    constexpr int D = M355_ZI_POISSON;
    const double psi = p[0], mu = p[1];
    const double m2 = mu * mu, m3 = m2 * mu;
    const double v = psi * (mu + (1.0 - psi) * m2);
    switch (s) {
      case M355_MEAN: return psi * mu;
      case M355_MODE: return zi_mode<D>(floor_(mu), p);
      case M355_MEDIAN: return dist_fn<D, M355_PPF>(0.5, p);
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: return skew_from_raw(psi * mu, psi * (mu + m2), psi * ((mu + 3.0 * m2) + m3));
      case M355_KURTOSIS:
        return kurt_from_raw(psi * mu, psi * (mu + m2), psi * ((mu + 3.0 * m2) + m3), psi * (((mu + 7.0 * m2) + 6.0 * m3) + m2 * m2));
      case M355_ENTROPY: return discrete_entropy<D, C>(0.0, (mu < 10.0) ? 25.0 : __builtin_trunc(mu * 3.0), p);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_ZI_BINOMIAL> {  // zi_binomial.py:9-72
  M355_MOM_HEAD(mbit(M355_ENTROPY)) {
    constexpr int D = M355_ZI_BINOMIAL;
    const double psi = p[0], n = p[1], q = p[2];
    const double np_ = n * q, qc = 1.0 - q;
    const double v = psi * (np_ * qc + (1.0 - psi) * (np_ * np_));
    switch (s) {
      case M355_MEAN: return psi * n * q;
      case M355_MODE: return zi_mode<D>(floor_((n + 1.0) * q), p);
      case M355_MEDIAN: return dist_fn<D, M355_PPF>(0.5, p);
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: {
        const double b2 = np_ * qc + np_ * np_;
        const double b3 = (np_ * qc * (1.0 - 2.0 * q) + 3.0 * (np_ * np_) * qc) + np_ * np_ * np_;
        return skew_from_raw(psi * np_, psi * b2, psi * b3);
      }
      case M355_KURTOSIS: {
        const double f1 = np_, f2 = n * (n - 1.0) * (q * q), f3 = n * (n - 1.0) * (n - 2.0) * (q * q * q),
                     f4 = n * (n - 1.0) * (n - 2.0) * (n - 3.0) * ((q * q) * (q * q));
        return kurt_from_raw(psi * f1, psi * (f2 + f1), psi * ((f3 + 3.0 * f2) + f1), psi * (((f4 + 6.0 * f3) + 7.0 * f2) + f1));
      }
      case M355_ENTROPY: return discrete_entropy<D, C>(0.0, n + 1.0, p);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_ZI_NEGATIVEBINOMIAL> {  // zi_negativebinomial.py:9-79
  M355_MOM_HEAD(mbit(M355_ENTROPY)) {
    constexpr int D = M355_ZI_NEGATIVEBINOMIAL;
    const double psi = p[0], n = p[1], q = p[2];
    const double qc = 1.0 - q;
    const double bm = n * qc / q;
    const double v = psi * (n * qc / (q * q) + (1.0 - psi) * (bm * bm));
    const double r = qc / q;
    const double f1 = n * qc / q, f2 = n * (n + 1.0) * (qc * qc) / (q * q), f3 = n * (n + 1.0) * (n + 2.0) * (qc * qc * qc) / (q * q * q),
                 f4 = n * (n + 1.0) * (n + 2.0) * (n + 3.0) * ((qc * qc) * (qc * qc)) / ((q * q) * (q * q));
    (void)r;
    switch (s) {
      case M355_MEAN: return psi * n * qc / q;
      case M355_MODE: return zi_mode<D>((n < 1.0) ? 0.0 : floor_((n - 1.0) * qc / q), p);
      case M355_MEDIAN: return dist_fn<D, M355_PPF>(0.5, p);
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: return skew_from_raw(psi * f1, psi * (f2 + f1), psi * ((f3 + 3.0 * f2) + f1));
      case M355_KURTOSIS: return kurt_from_raw(psi * f1, psi * (f2 + f1), psi * ((f3 + 3.0 * f2) + f1), psi * (((f4 + 6.0 * f3) + 7.0 * f2) + f1));
      case M355_ENTROPY: return discrete_entropy<D, C>(dist_fn<D, M355_PPF>(0.0001, p), dist_fn<D, M355_PPF>(0.9999, p), p);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_WALD> {  // wald.py:5-40
  M355_MOM_HEAD(0) {
    const double mu = p[0], lam = p[1];
    switch (s) {
      case M355_MEAN: return mu;
      case M355_MODE: {
        const double ratio = 3.0 * mu / (2.0 * lam);
        return mu * (sqrt_(1.0 + ratio * ratio) - ratio);
      }
      case M355_MEDIAN: return dist_fn<M355_WALD, M355_PPF>(0.5, p);
      case M355_VAR: return mu * mu * mu / lam;
      case M355_STD: return sqrt_(mu * mu * mu / lam);
      case M355_SKEWNESS: return 3.0 * sqrt_(mu / lam);
      case M355_KURTOSIS: return 15.0 * mu / lam;
      case M355_ENTROPY: {
        const double x = 2.0 * lam / mu;
        const double gamma_term = -gammainc_q(1e-8, x, lgamma_(1e-8)) * tgamma_(1e-8);
        return 0.5 * flog((2.0 * kPi * 2.718281828459045 * (mu * mu * mu)) / lam) + 1.5 * exp_(x) * gamma_term;
      }
    }
    return kNaN;
  }
};

template <> struct Mom<M355_EXGAUSSIAN> {  // exgaussian.py:8-50
  M355_MOM_HEAD(mbit(M355_ENTROPY)) {
    constexpr int D = M355_EXGAUSSIAN;
    const double mu = p[0], sg = p[1], nu = p[2];
    const double r = nu / sg, nus2 = r * r, op = 1.0 + nus2;
    switch (s) {
      case M355_MEAN: return mu + nu;
      case M355_MODE: {
        const double tau = 1.0 / nu;
        const double sgn = (tau > 0.0) ? 1.0 : ((tau < 0.0) ? -1.0 : tau);
        return (mu - sgn * 1.4142135623730951 * sg * erfcinv_(__builtin_fabs(tau) / sg * 0.79788456080286535588)) + sg * sg / tau;
      }
      case M355_MEDIAN: return dist_fn<D, M355_PPF>(0.5, p);
      case M355_VAR: return sg * sg + nu * nu;
      case M355_STD: return sqrt_(sg * sg + nu * nu);
      case M355_SKEWNESS: return 2.0 * (r * r * r) / (op * sqrt_(op));
      case M355_KURTOSIS: return 6.0 * (nus2 * nus2) / (op * op);
      case M355_ENTROPY: return continuous_entropy<D, C>(dist_fn<D, M355_PPF>(0.0001, p), dist_fn<D, M355_PPF>(0.9999, p), p);
    }
    return kNaN;
  }
};


// helper.py:375-422: grid point of linspace(lo, hi, 200) with the largest logpdf
template <int D, class C> M355_DEV double continuous_mode(double lo, double hi, const double* p) {
  const auto k = Dist<D>::derive(p);
  constexpr int n = 200;
  const double dx = (hi - lo) / (double)(n - 1);
  auto xat = [&](long long i) { return (i == n - 1) ? hi : lo + (double)i * dx; };
  return xat(C::argmax(n, [&](long long i) { return Dist<D>::template eval<M355_LOGPDF>(xat(i), k); }));
}

template <> struct Mom<M355_TRUNCATEDNORMAL> {  // truncatednormal.py:64-165
  M355_MOM_HEAD(0) {
    constexpr int D = M355_TRUNCATEDNORMAL;
    const double mu = p[0], sg = p[1];
    switch (s) {
      case M355_MODE: return (mu < p[2]) ? p[2] : ((mu > p[3]) ? p[3] : mu);
      case M355_MEDIAN: return dist_fn<D, M355_PPF>(0.5, p);
    }
    const auto k = Dist<D>::derive(p);
    const double al = k.alpha, be = k.beta, lz = k.logZ;
    const double ta = exp_((-0.5 * (al * al) - kLogSqrt2Pi) - lz), tb = exp_((-0.5 * (be * be) - kLogSqrt2Pi) - lz);
    const double var = sg * sg * ((1.0 + (al * ta - be * tb)) - (ta - tb) * (ta - tb));
    switch (s) {
      case M355_MEAN: return mu + sg * (ta - tb);
      case M355_VAR: return var;
      case M355_STD: return sqrt_(var);
      case M355_ENTROPY: return (flog(4.1327313541224930 * sg) + lz) + 0.5 * (al * ta - be * tb);  // sqrt(2 pi e)
    }
    const double m1 = ta - tb, m2 = (1.0 + al * ta) - be * tb;
    const double m3 = (2.0 * m1 + (al * al) * ta) - (be * be) * tb, m4 = (3.0 * m2 + (al * al * al) * ta) - (be * be * be) * tb;
    if (s == M355_SKEWNESS) return skew_from_raw(m1, m2, m3);
    if (s == M355_KURTOSIS) return kurt_from_raw(m1, m2, m3, m4);
    return kNaN;
  }
};

template <> struct Mom<M355_SKEW_STUDENTT> {  // skew_studentt.py:9-101
  static M355_DEV double eb(double a, double b) { return exp_(lbeta_(a, b)); }
  static M355_DEV double raw(int n, double a, double b) {  // skew_studentt.py:9-41
    const double coef = exp_((0.5 * n * flog(a + b) - n * 0.69314718055994530942) - lbeta_(a, b));
    double st;
    if (n == 1) st = eb(a + 0.5, b - 0.5) - eb(a - 0.5, b + 0.5);
    else if (n == 2) st = (eb(a + 1.0, b - 1.0) - 2.0 * eb(a, b)) + eb(a - 1.0, b + 1.0);
    else if (n == 3) st = ((eb(a + 1.5, b - 1.5) - 3.0 * eb(a + 0.5, b - 0.5)) + 3.0 * eb(a - 0.5, b + 0.5)) - eb(a - 1.5, b + 1.5);
    else st = (((eb(a + 2.0, b - 2.0) - 4.0 * eb(a + 1.0, b - 1.0)) + 6.0 * eb(a, b)) - 4.0 * eb(a - 1.0, b + 1.0)) + eb(a - 2.0, b + 2.0);
    return coef * st;
  }
  M355_MOM_HEAD(mbit(M355_MODE) | mbit(M355_ENTROPY)) {
    constexpr int D = M355_SKEW_STUDENTT;
    const double a = p[0], b = p[1], mu = p[2], sg = p[3];
    switch (s) {
      case M355_MEAN: return ((a > 0.5) & (b > 0.5)) ? mu + sg * raw(1, a, b) : kNaN;
      case M355_MODE: {
        const double sd = sg * sqrt_((a + b) / (2.0 * a * b));
        return continuous_mode<D, C>(mu - 5.0 * sd, mu + 5.0 * sd, p);
      }
      case M355_MEDIAN: return dist_fn<D, M355_PPF>(0.5, p);
      case M355_ENTROPY: return continuous_entropy<D, C>(mu - 50.0 * sg, mu + 50.0 * sg, p);
    }
    const double e1 = raw(1, a, b), e2 = raw(2, a, b);
    const double vz = e2 - e1 * e1;
    switch (s) {
      case M355_VAR: return ((a > 1.0) & (b > 1.0)) ? sg * sg * vz : kNaN;
      case M355_STD: {
        const double v = ((a > 1.0) & (b > 1.0)) ? sg * sg * vz : kNaN;
        return (v - v == 0.0) ? sqrt_(v) : v;
      }
      case M355_SKEWNESS: {
        const double e3 = raw(3, a, b), sz = sqrt_(vz);
        return ((a > 1.5) & (b > 1.5)) ? ((e3 - 3.0 * e1 * e2) + 2.0 * (e1 * e1 * e1)) / (sz * sz * sz) : kNaN;
      }
      case M355_KURTOSIS: {
        const double e3 = raw(3, a, b), e4 = raw(4, a, b);
        const double c4 = ((e4 - 4.0 * e1 * e3) + 6.0 * (e1 * e1) * e2) - 3.0 * ((e1 * e1) * (e1 * e1));
        return ((a > 2.0) & (b > 2.0)) ? c4 / (vz * vz) - 3.0 : kNaN;
      }
    }
    return kNaN;
  }
};


template <> struct Mom<M355_SKEWNORMAL> {  // skewnormal.py:8-59
  M355_MOM_HEAD(0) {
    const double mu = p[0], sg = p[1], al = p[2];
    const double delta = al / sqrt_(1.0 + al * al);
    const double mz = 0.79788456080286535588 * delta;  // sqrt(2 / pi) delta
    switch (s) {
      case M355_MEAN: return mu + sg * 0.79788456080286535588 * al / sqrt_(1.0 + al * al);
      case M355_MODE: {
        const double term2 = (1.0 - kPi / 4.0) * (mz * mz * mz) / (1.0 - (2.0 / kPi) * (delta * delta));
        const double sgn = (al > 0.0) ? 1.0 : ((al < 0.0) ? -1.0 : al);
        const double term3 = 0.5 * sgn * exp_(-2.0 * kPi / __builtin_fabs(al));
        return mu + sg * ((mz - term2) - term3);
      }
      case M355_MEDIAN: return dist_fn<M355_SKEWNORMAL, M355_PPF>(0.5, p);
      case M355_VAR: return sg * sg * (1.0 - 2.0 * (delta * delta) / kPi);
      case M355_STD: return sqrt_(sg * sg * (1.0 - 2.0 * (delta * delta) / kPi));
      case M355_SKEWNESS: {
        const double d = 1.0 - mz * mz;
        return ((4.0 - kPi) / 2.0) * ((mz * mz * mz) / (d * sqrt_(d)));
      }
      case M355_KURTOSIS: {
        const double vz = 1.0 - 2.0 * (delta * delta) / kPi;
        return 2.0 * (kPi - 3.0) * (((mz * mz) * (mz * mz)) / (vz * vz));
      }
      case M355_ENTROPY: {
        const double hn = 0.5 * flog(2.0 * kPi * 2.718281828459045 * (sg * sg)), hh = 0.5 * flog(kPi * (sg * sg) / 2.0) + 0.5;
        const double f = 1.0 / sqrt_(1.0 + al * al);
        return hn * f + hh * (1.0 - f);
      }
    }
    return kNaN;
  }
};

template <> struct Mom<M355_VONMISES> {  // vonmises.py:6-42
  M355_MOM_HEAD(0) {
    const double mu = p[0], kp = p[1];
    switch (s) {
      case M355_MEAN: case M355_MODE: case M355_MEDIAN: return mu;
      case M355_SKEWNESS: case M355_KURTOSIS: return 0.0;
    }
    const double i0 = ive01_(0, kp);
    const double v = 1.0 - ive01_(1, kp) / i0;
    switch (s) {
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_ENTROPY: return flog(2.0 * kPi * i0) + kp * v;
    }
    return kNaN;
  }
};

template <> struct Mom<M355_RICE> {  // rice.py:13-60, bounds rice.py:118-123
  M355_MOM_HEAD(mbit(M355_MODE) | mbit(M355_SKEWNESS) | mbit(M355_KURTOSIS) | mbit(M355_ENTROPY)) {
    constexpr int D = M355_RICE;
    const double nu = p[0], sg = p[1];
    const double lo = 1e-10, hi = nu + 10.0 * sg;
    switch (s) {
      case M355_MODE: return continuous_mode<D, C>(lo, hi, p);
      case M355_MEDIAN: return dist_fn<D, M355_PPF>(0.5, p);
      case M355_SKEWNESS: return continuous_skewness<D, C>(lo, hi, p);
      case M355_KURTOSIS: return continuous_kurtosis<D, C>(lo, hi, p);
      case M355_ENTROPY: return continuous_entropy<D, C>(lo, hi, p);
    }
    const double q = nu * nu / (2.0 * (sg * sg));
    const double lh = (1.0 + q) * ive01_(0, q / 2.0) + q * ive01_(1, q / 2.0);
    const double v = (2.0 * (sg * sg) + nu * nu) - (kPi * (sg * sg) / 2.0) * (lh * lh);
    switch (s) {
      case M355_MEAN: return sg * sqrt_(kPi / 2.0) * lh;
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
    }
    return kNaN;
  }
};

template <> struct Mom<M355_POLYAGAMMA> {  // polyagamma.py:15-111
  M355_MOM_HEAD(mbit(M355_MODE) | mbit(M355_SKEWNESS) | mbit(M355_KURTOSIS) | mbit(M355_ENTROPY)) {
    constexpr int D = M355_POLYAGAMMA;
    const double h = p[0], z = p[1];
    const bool small = __builtin_fabs(z) < 1e-6;
    const double sz = small ? 1.0 : z;
    const double ch = ::cosh(sz / 2.0);
    const double m = small ? h / 4.0 : h / (2.0 * sz) * ::tanh(sz / 2.0);
    const double v = small ? h / 24.0 : h / (4.0 * (sz * sz * sz)) * (::sinh(sz) - sz) / (ch * ch);
    const double lo = 1e-10, hi = m + 10.0 * sqrt_(v);
    switch (s) {
      case M355_MEAN: return m;
      case M355_MODE: return continuous_mode<D, C>(lo, hi, p);
      case M355_MEDIAN: return dist_fn<D, M355_PPF>(0.5, p);
      case M355_VAR: return v;
      case M355_STD: return sqrt_(v);
      case M355_SKEWNESS: return continuous_skewness<D, C>(lo, hi, p);
      case M355_KURTOSIS: return continuous_kurtosis<D, C>(lo, hi, p);
      case M355_ENTROPY: return continuous_entropy<D, C>(lo, hi, p);
    }
    return kNaN;
  }
};

}  // namespace m355
