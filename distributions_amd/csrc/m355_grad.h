// Gradient path: d logpdf / d(x, params) by forward-mode dual numbers whose
// set of live partials is a COMPILE-TIME mask (bit 0 = x, bit 1 + j = parameter j).  An expression
// that does not depend on a variable carries no slot for it, so (a) no multiply-by-zero work is
// generated and (b) an infinite factor on some other path cannot turn a partial into NaN -- the
// behaviour of reverse-mode differentiation of the reference's graph (pytensor.grad), which is what
// a PyMC-style caller of the reference gets.  Derivative rules of the primitives follow PyTensor's
// (log, log1p, exp, gammaln -> digamma, pow -> [y x^(y-1), log(x) x^y], abs -> sgn,
// xlogy0(a, y) -> [log y, a / y], switch -> the selected arm).  Each Grad<D> restates the logpdf
// graph of the reference module (file:line cited; paths relative to
// /root/reference/pytensor_distributions) split in two: pre() = everything that depends on the
// parameters only (evaluated once per thread / per element of a parameter table), lp() = the
// per-element part.  (The CPU restatement it is tested against lives with the tests, not here.)
#pragma once
#include <utility>

#include "m355_dists3.h"

namespace m355 {

// ------------------------------------------------------------------------------------- digamma
// psi(x): recurrence up to x >= 10, then the asymptotic series (DLMF 5.11.2) to x^-14; reflection
// (DLMF 5.5.4) for x < 0.  |error| <= 2e-15 max(1, |psi|) for x > 0 (tests/test_hostmath.py).
M355_HD double digamma_(double x) {
  if (!(x == x) || x == -kInf) return kNaN;
  if (x == kInf) return kInf;
  double refl = 0.0;
  if (x <= 0.0) {
    if (x == 0.0) return __builtin_copysign(kInf, -x);  // -inf at +0, +inf at -0 (SciPy's convention)
    if (x == __builtin_floor(x)) return kNaN;          // the other poles
    refl = -kPi / ::tan(kPi * (x - __builtin_rint(x)));  // period 1: reduce exactly before scaling by pi
    x = 1.0 - x;
  }
  double acc = 0.0;
  while (x < 10.0) {
    acc -= 1.0 / x;
    x += 1.0;
  }
  const double r = 1.0 / x, r2 = r * r;
  // B_2k / (2k): 1/12, -1/120, 1/252, -1/240, 1/132, -691/32760, 1/12
  double s = 1.0 / 12.0;
  s = __builtin_fma(s, r2, -691.0 / 32760.0);
  s = __builtin_fma(s, r2, 1.0 / 132.0);
  s = __builtin_fma(s, r2, -1.0 / 240.0);
  s = __builtin_fma(s, r2, 1.0 / 252.0);
  s = __builtin_fma(s, r2, -1.0 / 120.0);
  s = __builtin_fma(s, r2, 1.0 / 12.0);
  return ((flog(x) - 0.5 * r) - s * r2) + acc + refl;
}

// ---------------------------------------------------------------------------------- dual numbers
constexpr int kGradSlots = 1 + 4;  // x + kMaxParams
constexpr int popc(unsigned m) { return __builtin_popcount(m); }

template <unsigned M> struct Dual {
  static constexpr unsigned mask = M;
  double v;
  double d[popc(M) > 0 ? popc(M) : 1];
  template <int B> M355_HD double get() const {
    if constexpr ((M >> B) & 1u) return d[popc(M & ((1u << B) - 1u))];
    else return 0.0;
  }
  template <int B> M355_HD void set(double t) {
    if constexpr ((M >> B) & 1u) d[popc(M & ((1u << B) - 1u))] = t;
  }
};
template <class T> struct is_dual { static constexpr bool value = false; };
template <unsigned M> struct is_dual<Dual<M>> { static constexpr bool value = true; };

M355_HD double val(double a) { return a; }
template <unsigned M> M355_HD double val(const Dual<M>& a) { return a.v; }

template <int I, unsigned A, unsigned B>
M355_HD double comb1(double fa, const Dual<A>& a, double fb, const Dual<B>& b) {
  if constexpr (((A >> I) & 1u) && ((B >> I) & 1u)) return fa * a.template get<I>() + fb * b.template get<I>();
  else if constexpr ((A >> I) & 1u) return fa * a.template get<I>();
  else if constexpr ((B >> I) & 1u) return fb * b.template get<I>();
  else return 0.0;
}
template <unsigned A, unsigned B, int... I>
M355_HD Dual<(A | B)> comb_impl(double v, double fa, const Dual<A>& a, double fb, const Dual<B>& b,
                                std::integer_sequence<int, I...>) {
  Dual<(A | B)> r;
  r.v = v;
  (r.template set<I>(comb1<I>(fa, a, fb, b)), ...);
  return r;
}
// value v with partials fa * a.d + fb * b.d over the union of the live slots
template <unsigned A, unsigned B>
M355_HD Dual<(A | B)> comb(double v, double fa, const Dual<A>& a, double fb, const Dual<B>& b) {
  return comb_impl(v, fa, a, fb, b, std::make_integer_sequence<int, kGradSlots>{});
}
template <unsigned A> M355_HD Dual<A> chain(double v, double f, const Dual<A>& a) {
  return comb(v, f, a, 0.0, Dual<0>{0.0, {0.0}});
}

template <unsigned A, unsigned B> M355_HD auto operator+(const Dual<A>& a, const Dual<B>& b) { return comb(a.v + b.v, 1.0, a, 1.0, b); }
template <unsigned A> M355_HD auto operator+(const Dual<A>& a, double b) { return chain(a.v + b, 1.0, a); }
template <unsigned A> M355_HD auto operator+(double b, const Dual<A>& a) { return chain(b + a.v, 1.0, a); }
template <unsigned A, unsigned B> M355_HD auto operator-(const Dual<A>& a, const Dual<B>& b) { return comb(a.v - b.v, 1.0, a, -1.0, b); }
template <unsigned A> M355_HD auto operator-(const Dual<A>& a, double b) { return chain(a.v - b, 1.0, a); }
template <unsigned A> M355_HD auto operator-(double b, const Dual<A>& a) { return chain(b - a.v, -1.0, a); }
template <unsigned A> M355_HD auto operator-(const Dual<A>& a) { return chain(-a.v, -1.0, a); }
template <unsigned A, unsigned B> M355_HD auto operator*(const Dual<A>& a, const Dual<B>& b) { return comb(a.v * b.v, b.v, a, a.v, b); }
template <unsigned A> M355_HD auto operator*(const Dual<A>& a, double b) { return chain(a.v * b, b, a); }
template <unsigned A> M355_HD auto operator*(double b, const Dual<A>& a) { return chain(b * a.v, b, a); }
template <unsigned A, unsigned B> M355_HD auto operator/(const Dual<A>& a, const Dual<B>& b) {
  const double r = 1.0 / b.v, q = a.v * r;
  return comb(a.v / b.v, r, a, -(q * r), b);
}
template <unsigned A> M355_HD auto operator/(const Dual<A>& a, double b) { return chain(a.v / b, 1.0 / b, a); }
template <unsigned A> M355_HD auto operator/(double b, const Dual<A>& a) {
  const double r = 1.0 / a.v, q = b * r;
  return chain(b / a.v, -(q * r), a);
}

M355_HD double dlog(double a) { return flog(a); }
M355_HD double dlog1p(double a) { return flog1p(a); }
M355_HD double dsqr(double a) { return a * a; }
template <unsigned A> M355_HD auto dlog(const Dual<A>& a) { return chain(flog_tab(a.v), 1.0 / a.v, a); }
template <unsigned A> M355_HD auto dlog1p(const Dual<A>& a) { return chain(flog1p_tab(a.v), 1.0 / (1.0 + a.v), a); }
template <unsigned A> M355_DEV auto dexp(const Dual<A>& a) {
  const double e = exp_(a.v);
  return chain(e, e, a);
}
template <unsigned A> M355_HD auto dlgamma(const Dual<A>& a) { return chain(lgamma_(a.v), digamma_(a.v), a); }
template <unsigned A> M355_HD auto dsqr(const Dual<A>& a) { return chain(a.v * a.v, 2.0 * a.v, a); }
template <unsigned A> M355_HD auto dabs(const Dual<A>& a) {
  return chain(__builtin_fabs(a.v), a.v > 0.0 ? 1.0 : (a.v < 0.0 ? -1.0 : 0.0), a);
}
// PyTensor Pow.grad: [gz * y * x**(y-1), gz * log(x) * x**y]
template <unsigned A, unsigned B> M355_DEV auto dpow(const Dual<A>& a, const Dual<B>& b) {
  const double v = pow_(a.v, b.v);
  const double da = (a.v == 0.0 || !(v - v == 0.0)) ? b.v * pow_(a.v, b.v - 1.0) : v * b.v / a.v;
  return comb(v, da, a, flog(a.v) * v, b);
}
// pytensor.tensor.xlogx.XlogY0: value 0 where a == 0; grad [gz * log(y), gz * a / y]
template <unsigned A, unsigned B> M355_HD auto dxlogy0(const Dual<A>& a, const Dual<B>& y) {
  const double ly = flog_tab(y.v);
  return comb(a.v == 0.0 ? 0.0 : a.v * ly, ly, a, a.v / y.v, y);
}
// pt.switch as a select: value and partials of the selected arm (real selects: the dead arm's NaNs stay dead)
template <int I, unsigned A, unsigned B> M355_HD double sel1(bool c, const Dual<A>& a, const Dual<B>& b) {
  return c ? a.template get<I>() : b.template get<I>();
}
template <unsigned A, unsigned B, int... I>
M355_HD Dual<(A | B)> dselect_impl(bool c, const Dual<A>& a, const Dual<B>& b, std::integer_sequence<int, I...>) {
  Dual<(A | B)> r;
  r.v = c ? a.v : b.v;
  (r.template set<I>(sel1<I>(c, a, b)), ...);
  return r;
}
template <unsigned A, unsigned B> M355_HD auto dselect(bool c, const Dual<A>& a, const Dual<B>& b) {
  return dselect_impl(c, a, b, std::make_integer_sequence<int, kGradSlots>{});
}
// the same value with (structurally zero) slots added, so that two arms of a branch share a type
template <unsigned M, unsigned A> M355_HD Dual<(M | A)> widen(const Dual<A>& a) {
  Dual<M> z;
  z.v = 0.0;
#pragma unroll
  for (int i = 0; i < (popc(M) > 0 ? popc(M) : 1); ++i) z.d[i] = 0.0;
  return dselect(true, a, z);
}
// the -inf arm of a support switch: value -inf, every partial exactly 0 (0 * inf must not appear)
template <unsigned B> M355_HD Dual<B> outside(bool c, const Dual<B>& b) {
  Dual<B> r;
  r.v = c ? -kInf : b.v;
#pragma unroll
  for (int i = 0; i < (popc(B) > 0 ? popc(B) : 1); ++i) r.d[i] = c ? 0.0 : b.d[i];
  return r;
}
template <unsigned A, unsigned B> M355_HD auto dbetaln(const Dual<A>& a, const Dual<B>& b) {
  return (dlgamma(a) + dlgamma(b)) - dlgamma(a + b);
}
// normal.logcdf on a standardised z (normal.py:70-76); d/dz = phi(z)/Phi(z) = exp(-z^2/2 - log sqrt(2 pi) - logcdf)
template <unsigned A> M355_DEV auto dnormal_logcdf_z(const Dual<A>& z) {
  const double v = normal_logcdf_z(z.v);
  return chain(v, exp_((-0.5 * z.v * z.v - kLogSqrt2Pi) - v), z);
}

template <class A, class B, class C = double, class D = double, class E = double, class F = double> struct Pack {
  A a;
  B b;
  C c;
  D d;
  E e;
  F f;
};
template <class A, class B> M355_HD Pack<A, B> pack(A a, B b) { return {a, b, 0.0, 0.0, 0.0, 0.0}; }
template <class A, class B, class C> M355_HD Pack<A, B, C> pack(A a, B b, C c) { return {a, b, c, 0.0, 0.0, 0.0}; }
template <class A, class B, class C, class D> M355_HD Pack<A, B, C, D> pack(A a, B b, C c, D d) { return {a, b, c, d, 0.0, 0.0}; }
template <class A, class B, class C, class D, class E> M355_HD Pack<A, B, C, D, E> pack(A a, B b, C c, D d, E e) {
  return {a, b, c, d, e, 0.0};
}
template <class A, class B, class C, class D, class E, class F> M355_HD Pack<A, B, C, D, E, F> pack(A a, B b, C c, D d, E e, F f) {
  return {a, b, c, d, e, f};
}

// --------------------------------------------------------------------------- logpdf graphs
template <int D> struct Grad { static constexpr bool kHas = false; static constexpr int NP = 0; };
#define M355_GRAD_HEAD(np) static constexpr bool kHas = true; static constexpr int NP = np;

template <> struct Grad<M355_NORMAL> {  // normal.py:79-80
  M355_GRAD_HEAD(2)
  template <class M, class S> static M355_DEV auto pre(M mu, S sigma) { return pack(mu, 1.0 / sigma, dlog(sigma) + kLogSqrt2Pi); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) { return -0.5 * dsqr((x - k.a) * k.b) - k.c; }
};

template <> struct Grad<M355_GAMMA> {  // gamma.py:77-82 (quirk Q2: xlogy0)
  M355_GRAD_HEAD(2)
  template <class A, class B> static M355_DEV auto pre(A alpha, B beta) {
    return pack(alpha - 1.0, beta, alpha * dlog(beta) - dlgamma(alpha));
  }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    return outside(val(x) < 0.0, (dxlogy0(k.a, x) - k.b * x) + k.c);
  }
  // x as data, scalar parameters: sum_i lp(x_i) = a * sum log x_i - b * sum x_i + n * c is linear in the
  // sufficient statistics, so the dual-number expression is evaluated once per thread on the statistics
  // (sum of logarithms as a LogProd, m355_math.h) instead of once per element.  Elements outside
  // [min normal, inf) go through lp() itself (stats_add returns false).
  struct Stats {
    LogProd px;
    double sx = 0.0;
    int cnt = 0;
  };
  static M355_DEV bool stats_ok(const double* p) { return __builtin_fabs(p[0] - 1.0) <= kSumAccMaxCoef; }
  template <class K> static M355_DEV bool stats_add(Stats& s, double x, const K&) {
    if (!((x >= kMinNormal) & (x < kInf))) return false;
    s.px.mul(x);
    s.sx += x;
    ++s.cnt;
    return true;
  }
  static M355_DEV void stats_renorm(Stats& s) { s.px.renorm(); }
  template <class K> static M355_DEV auto stats_lp(const Stats& s, const K& k) {
    return (k.a * s.px.log() - k.b * s.sx) + k.c * (double)s.cnt;
  }
};

template <> struct Grad<M355_BETA> {  // beta.py:102-108
  M355_GRAD_HEAD(2)
  template <class A, class B> static M355_DEV auto pre(A a, B b) { return pack(a - 1.0, b - 1.0, dbetaln(a, b)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    return outside((val(x) < 0.0) || (val(x) > 1.0), (dxlogy0(k.a, x) + dxlogy0(k.b, 1.0 - x)) - k.c);
  }
  // sufficient statistics sum log x, sum log(1 - x) (see Grad<M355_GAMMA>::Stats)
  struct Stats {
    LogProd px, pc;
    int cnt = 0;
  };
  static M355_DEV bool stats_ok(const double* p) {
    return (__builtin_fabs(p[0] - 1.0) <= kSumAccMaxCoef) & (__builtin_fabs(p[1] - 1.0) <= kSumAccMaxCoef);
  }
  template <class K> static M355_DEV bool stats_add(Stats& s, double x, const K&) {
    if (!((x >= kMinNormal) & (x < 1.0))) return false;
    s.px.mul(x);
    s.pc.mul(1.0 - x);
    ++s.cnt;
    return true;
  }
  static M355_DEV void stats_renorm(Stats& s) {
    s.px.renorm();
    s.pc.renorm();
  }
  template <class K> static M355_DEV auto stats_lp(const Stats& s, const K& k) {
    return (k.a * s.px.log() + k.b * s.pc.log()) - k.c * (double)s.cnt;
  }
};

template <> struct Grad<M355_STUDENTT> {  // studentt.py:108-119 (quirk Q1: the nu > 1e5 arm is normal.logcdf)
  M355_GRAD_HEAD(3)
  template <class N, class M, class S> static M355_DEV auto pre(N nu, M mu, S sigma) {
    const auto c = (dlgamma((nu + 1.0) / 2.0) - dlgamma(nu / 2.0)) - 0.5 * dlog(nu * kPi * dsqr(sigma));
    return pack(nu, mu, 1.0 / sigma, c, 0.5 * (nu + 1.0), 1.0 / (dsqr(sigma) * nu));
  }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    if (val(k.a) > 1e5) return widen<0xFu>(dnormal_logcdf_z((x - k.b) * k.c));
    return widen<0xFu>(k.d - k.e * dlog1p(dsqr(x - k.b) * k.f));
  }
  // x as data, scalar parameters: sum_i lp = n d - e T with T = sum log1p(w_i), w = (x - mu)^2 f.  T's value is a
  // LogProd (+ the first-order rounding of 1 + w, as flog1p_tab); its partials are dT/dmu = -2 f sum delta/(1+w)
  // and dT/df = sum delta^2/(1+w): one division per element instead of a logarithm and a division.
  struct Stats {
    LogProd pu;
    double corr = 0.0, s1 = 0.0, s2 = 0.0;
    int cnt = 0;
  };
  static M355_DEV bool stats_ok(const double* p) { return (0.5 * (p[0] + 1.0) <= kSumAccMaxCoef) & !(p[0] > 1e5); }
  template <class K> static M355_DEV bool stats_add(Stats& s, double x, const K& k) {
    const double d = x - val(k.b);
    const double w = d * d * val(k.f);
    if (!(w < 0x1p1000)) return false;
    const double u = 1.0 + w;
    const double c = w - (u - 1.0);
    s.corr += (w < 0.5) ? c * (2.0 - u) : 0.0;
    s.pu.mul(u);
    const double r = 1.0 / u;
    const double dr = d * r;
    s.s1 += dr;
    s.s2 += d * dr;
    ++s.cnt;
    return true;
  }
  static M355_DEV void stats_renorm(Stats& s) { s.pu.renorm(); }
  template <class K> static M355_DEV auto stats_lp(const Stats& s, const K& k) {
    const auto t = comb(s.pu.log() + s.corr, -2.0 * val(k.f) * s.s1, k.b, s.s2, k.f);
    return widen<0xEu>(k.d * (double)s.cnt - k.e * t);
  }
};

template <> struct Grad<M355_POISSON> {  // poisson.py:70-71
  M355_GRAD_HEAD(1)
  template <class M> static M355_DEV auto pre(M mu) { return pack(mu, 0.0); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) { return (dxlogy0(x, k.a) - dlgamma(x + 1.0)) - k.a; }
};

template <> struct Grad<M355_BINOMIAL> {  // binomial.py:66-75
  M355_GRAD_HEAD(2)
  template <class N, class P> static M355_DEV auto pre(N n, P p) { return pack(n, p, dlgamma(n + 1.0), dlog1p(-p)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    const auto nmx = k.a - x;
    const auto r = (((k.c - dlgamma(x + 1.0)) - dlgamma(nmx + 1.0)) + dxlogy0(x, k.b)) + nmx * k.d;
    return outside((val(x) < 0.0) || (val(x) > val(k.a)), r);
  }
};

template <> struct Grad<M355_EXPONENTIAL> {  // exponential.py:71-72
  M355_GRAD_HEAD(1)
  template <class L> static M355_DEV auto pre(L lam) { return pack(lam, dlog(lam)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) { return outside(val(x) < 0.0, k.b - k.a * x); }
};

template <> struct Grad<M355_LOGNORMAL> {  // lognormal.py:83-91
  M355_GRAD_HEAD(2)
  template <class M, class S> static M355_DEV auto pre(M mu, S sigma) { return pack(mu, 1.0 / sigma, dlog(sigma) + kHalfLog2Pi); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    const auto lx = dlog(x);
    return outside(val(x) <= 0.0, ((-lx) - k.c) - 0.5 * dsqr((lx - k.a) * k.b));
  }
};

template <> struct Grad<M355_HALFNORMAL> {  // halfnormal.py:67-70
  M355_GRAD_HEAD(1)
  template <class S> static M355_DEV auto pre(S sigma) { return pack(1.0 / sigma, -0.22579135264472743236 - dlog(sigma)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) { return outside(val(x) < 0.0, k.b - 0.5 * dsqr(x * k.a)); }
};

template <> struct Grad<M355_CAUCHY> {  // cauchy.py:72-73
  M355_GRAD_HEAD(2)
  template <class A, class B> static M355_DEV auto pre(A alpha, B beta) { return pack(alpha, 1.0 / beta, -kLogPi - dlog(beta)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) { return k.c - dlog(1.0 + dsqr((x - k.a) * k.b)); }
};

template <> struct Grad<M355_HALFCAUCHY> {  // halfcauchy.py:72-75
  M355_GRAD_HEAD(1)
  template <class B> static M355_DEV auto pre(B beta) { return pack(1.0 / beta, kLog2 - dlog(kPi * beta)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) { return outside(val(x) < 0.0, k.b - dlog(1.0 + dsqr(x * k.a))); }
};

template <> struct Grad<M355_LAPLACE> {  // laplace.py:72-73
  M355_GRAD_HEAD(2)
  template <class M, class B> static M355_DEV auto pre(M mu, B b) { return pack(mu, 1.0 / b, dlog(b)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) { return (kLogHalf - dabs((x - k.a) * k.b)) - k.c; }
};

template <> struct Grad<M355_LOGISTIC> {  // logistic.py:74-76
  M355_GRAD_HEAD(2)
  template <class M, class S> static M355_DEV auto pre(M mu, S s) { return pack(mu, 1.0 / s, dlog(s)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    const auto z = (x - k.a) * k.b;
    return outside(val(x) == -kInf, ((-k.c) - z) - 2.0 * dlog1p(dexp(-z)));
  }
};

template <> struct Grad<M355_GUMBEL> {  // gumbel.py:74-76
  M355_GRAD_HEAD(2)
  template <class M, class B> static M355_DEV auto pre(M mu, B beta) { return pack(mu, 1.0 / beta, dlog(beta)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    const auto z = (x - k.a) * k.b;
    return ((-z) - dexp(-z)) - k.c;
  }
};

template <> struct Grad<M355_WEIBULL> {  // weibull.py:82-87
  M355_GRAD_HEAD(2)
  template <class A, class B> static M355_DEV auto pre(A alpha, B beta) { return pack(alpha, 1.0 / beta, dlog(alpha / beta), alpha - 1.0); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    const auto z = x * k.b;
    return outside(val(x) < 0.0, (k.c + dxlogy0(k.d, z)) - dpow(z, k.a));
  }
};

template <> struct Grad<M355_PARETO> {  // pareto.py:77-80
  M355_GRAD_HEAD(2)
  template <class A, class M> static M355_DEV auto pre(A alpha, M m) { return pack(alpha + 1.0, m, dlog(alpha) + alpha * dlog(m)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) { return outside(val(x) < val(k.b), k.c - k.a * dlog(x)); }
};

template <> struct Grad<M355_UNIFORM> {  // uniform.py:78-80 (the selected arm does not contain x)
  M355_GRAD_HEAD(2)
  template <class L, class U> static M355_DEV auto pre(L lower, U upper) { return pack(lower, upper, -dlog(upper - lower)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    return outside((val(x) < val(k.a)) || (val(x) > val(k.b)), k.c + 0.0 * x);
  }
};

template <> struct Grad<M355_TRIANGULAR> {  // triangular.py:91-97
  M355_GRAD_HEAD(3)
  template <class L, class C, class U> static M355_DEV auto pre(L lower, C c, U upper) {
    return pack(lower, c, upper, 2.0 / ((upper - lower) * (c - lower)), 2.0 / ((upper - lower) * (upper - c)));
  }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    const auto res = dselect(val(x) < val(k.b), dlog((x - k.a) * k.d), dlog((k.c - x) * k.e));
    return outside(!((val(k.a) <= val(x)) && (val(x) <= val(k.c))), res);
  }
};

template <> struct Grad<M355_KUMARASWAMY> {  // kumaraswamy.py:80-85
  M355_GRAD_HEAD(2)
  template <class A, class B> static M355_DEV auto pre(A a, B b) { return pack(a, a - 1.0, b - 1.0, dlog(a * b)); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    return outside((val(x) <= 0.0) || (val(x) >= 1.0), (k.d + dxlogy0(k.b, x)) + dxlogy0(k.c, 1.0 - dpow(x, k.a)));
  }
};

template <> struct Grad<M355_MOYAL> {  // moyal.py:75-77
  M355_GRAD_HEAD(2)
  template <class M, class S> static M355_DEV auto pre(M mu, S sigma) { return pack(mu, 1.0 / sigma, dlog(sigma) + kHalfLog2Pi); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    const auto z = (x - k.a) * k.b;
    return -0.5 * (z + dexp(-z)) - k.c;
  }
};

template <> struct Grad<M355_ASYMMETRICLAPLACE> {  // asymmetriclaplace.py:88-95
  M355_GRAD_HEAD(3)
  template <class M, class B, class KP> static M355_DEV auto pre(M mu, B b, KP kappa) {
    const auto kinv = 1.0 / kappa;
    return pack(mu, 1.0 / b, -kappa, kinv, dlog(kappa + kinv) + dlog(b));
  }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    const auto xn = (x - k.a) * k.b;
    return dselect(val(xn) >= 0.0, xn * k.c, xn * k.d) - k.e;
  }
};

template <> struct Grad<M355_LOGLOGISTIC> {  // loglogistic.py:70-73
  M355_GRAD_HEAD(2)
  template <class A, class B> static M355_DEV auto pre(A alpha, B beta) { return pack(1.0 / alpha, beta, dlog(beta) - dlog(alpha), beta - 1.0); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    const auto z = x * k.a;
    return outside((val(x) <= 0.0) || (val(x) == kInf), (k.c + k.d * dlog(z)) - 2.0 * dlog1p(dpow(z, k.b)));
  }
};

template <> struct Grad<M355_LOGITNORMAL> {  // logitnormal.py:122-132 (logit = log(x) - log1p(-x), :11-12)
  M355_GRAD_HEAD(2)
  template <class M, class S> static M355_DEV auto pre(M mu, S sigma) { return pack(mu, 1.0 / sigma, dlog(sigma) + kHalfLog2Pi); }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    const auto lx = dlog(x);
    const auto l1 = dlog1p(-x);
    const auto z = ((lx - l1) - k.a) * k.b;
    return outside((val(x) <= 0.0) || (val(x) >= 1.0), ((-0.5 * dsqr(z) - k.c) - lx) - l1);
  }
};

template <> struct Grad<M355_INVERSEGAMMA> {  // inversegamma.py:76-81
  M355_GRAD_HEAD(2)
  template <class A, class B> static M355_DEV auto pre(A alpha, B beta) {
    return pack(alpha + 1.0, beta, alpha * dlog(beta) - dlgamma(alpha));
  }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    return outside(val(x) <= 0.0, (k.c - k.a * dlog(x)) - k.b / x);
  }
};

template <> struct Grad<M355_CHISQUARED> {  // chisquared.py:69-74
  M355_GRAD_HEAD(1)
  template <class N> static M355_DEV auto pre(N nu) {
    const auto h = nu / 2.0;
    return pack(h - 1.0, dlgamma(h) + (nu * kLog2) / 2.0);
  }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    return outside(val(x) < 0.0, (dxlogy0(k.a, x) - x / 2.0) - k.b);
  }
};

template <> struct Grad<M355_SCALEDINVERSECHISQUARED> {  // scaledinversechisquared.py:75-85
  M355_GRAD_HEAD(2)
  template <class N, class T> static M355_DEV auto pre(N nu, T tau2) {
    const auto h = nu / 2.0;
    return pack(h + 1.0, h * tau2, (dlog(tau2) * h - dlgamma(h)) + dlog(h) * h);
  }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    return outside(val(x) <= 0.0, ((-(dlog(x) * k.a)) - k.b / x) + k.c);
  }
};

template <> struct Grad<M355_HALFSTUDENTT> {  // halfstudentt.py:117-131
  M355_GRAD_HEAD(2)
  template <class N, class S> static M355_DEV auto pre(N nu, S sigma) {
    const auto c = ((dlgamma((nu + 1.0) / 2.0) - dlgamma(nu / 2.0)) - 0.5 * dlog(nu * kPi * dsqr(sigma))) + kLog2;
    return pack(nu, 1.0 / sigma, c, 0.5 * (nu + 1.0), 1.0 / nu, -0.22579135264472743236 - dlog(sigma));
  }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    const auto xn = x * k.b;
    if (val(k.a) > 1e5) return outside(val(x) < 0.0, widen<0x7u>(k.f - 0.5 * dsqr(xn)));
    return outside(val(x) < 0.0, widen<0x7u>(k.c - k.d * dlog(1.0 + dsqr(xn) * k.e)));
  }
};

template <> struct Grad<M355_BETASCALED> {  // betascaled.py:106-112
  M355_GRAD_HEAD(4)
  template <class A, class B, class L, class U> static M355_DEV auto pre(A a, B b, L lower, U upper) {
    return pack(a - 1.0, b - 1.0, lower, upper, dxlogy0((a + b) - 1.0, upper - lower) + dbetaln(a, b));
  }
  template <class X, class K> static M355_DEV auto lp(X x, const K& k) {
    return outside((val(x) < val(k.c)) || (val(x) > val(k.d)), (dxlogy0(k.a, x - k.c) + dxlogy0(k.b, k.d - x)) - k.e);
  }
};

constexpr bool has_grad(int dist) {
  return dist <= M355_BINOMIAL || (dist >= M355_EXPONENTIAL && dist <= M355_BETASCALED);
}

// parameters as dual variables (bit 1 + j), x as bit 0
template <class G, int... J> M355_DEV auto grad_pre(const double* p, std::integer_sequence<int, J...>) {
  return G::pre(Dual<(2u << J)>{p[J], {1.0}}...);
}
template <class G> M355_DEV auto grad_pre(const double* p) { return grad_pre<G>(p, std::make_integer_sequence<int, G::NP>{}); }

// out[0] = logpdf, out[1] = d/dx, out[2 + j] = d/dparam_j
template <class R, int... I> M355_DEV void grad_unpack(const R& r, double* out, std::integer_sequence<int, I...>) {
  out[0] = r.v;
  ((out[1 + I] = r.template get<I>()), ...);
}
// XM = 1: x is a variable (d/dx computed); XM = 0: x is data only -- every x-partial disappears at compile time
template <class G, unsigned XM = 1u, class K> M355_DEV void grad_eval(double x, const K& k, double* out) {
  const auto r = G::lp(Dual<XM>{x, {1.0}}, k);
  grad_unpack(r, out, std::make_integer_sequence<int, 1 + G::NP>{});
}

}  // namespace m355
