// Special functions the reference reaches through PyTensor scalar ops:
// gammainc / gammaincc / gammaincinv, betainc / betaincinv, and thin wrappers over the device
// libm for erf/erfc/erfcx/erfinv/lgamma/exp.  Written from the published series and continued
// fractions (DLMF 8.11.4, 8.9.2, 8.17.22; Stirling's series DLMF 5.11.1); every loop is bounded.
// All functions are per-lane scalar code: lanes of a wavefront that need different trip counts
// simply run masked, there is no cross-lane communication.
#pragma once
#include "m355_logfact128.h"
#include "m355_math.h"
#include "m355_erfcoef.h"
#include "m355_temmecoef.h"

namespace m355 {

constexpr double kPi = 3.14159265358979323846;
constexpr double kEps = 1.1102230246251565e-16;  // 2^-53
constexpr double kEuler = 0.57721566490153286061;
constexpr double kMaxLog = 709.782712893384;
constexpr double kMinLog = -745.13321910194122;

#if defined(__HIPCC__)
M355_HD double exp_(double x) { return ::exp(x); }
M355_DEV double expm1_(double x) { return ::expm1(x); }
M355_HD_HEAVY double lgamma_(double x) { return ::lgamma(x); }
M355_DEV double erf_(double x) { return ::erf(x); }
M355_DEV double erfc_(double x) { return ::erfc(x); }
M355_DEV double sqrt_(double x) { return __builtin_sqrt(x); }
M355_DEV double floor_(double x) { return __builtin_floor(x); }
// x**y: exp(y log x) on the common path (0 < x < inf, normal): relative error |y ln x| * 3e-16, i.e.
// <= 2e-13 even next to overflow and ~1e-15 typically, at a third of the cost of ocml's correctly
// rounded pow; every special case (zero, negative, inf, NaN, subnormal base) goes to ocml.
M355_DEV double pow_(double x, double y) {
  const unsigned hx = (unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32);
  if ((hx - 0x00100000u) < 0x7fe00000u) return ::exp(y * flog_pos(x));
  return ::pow(x, y);
}
M355_DEV double atan_(double x) { return ::atan(x); }
M355_DEV double tan_(double x) { return ::tan(x); }
// x**y where the caller already holds lx = log(x) (bodies that need both log(x) and x**y: weibull, loglogistic,
// kumaraswamy): one exp on the common path, ocml's pow for the special cases as in pow_.
M355_DEV double pow_l(double x, double lx, double y) {
  const unsigned hx = (unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32);
  if ((hx - 0x00100000u) < 0x7fe00000u) return ::exp(y * lx);
  return ::pow(x, y);
}
#else
M355_DEV double exp_(double x) { return std::exp(x); }
M355_DEV double expm1_(double x) { return std::expm1(x); }
M355_DEV double lgamma_(double x) { return std::lgamma(x); }
M355_DEV double erf_(double x) { return std::erf(x); }
M355_DEV double erfc_(double x) { return std::erfc(x); }
M355_DEV double sqrt_(double x) { return std::sqrt(x); }
M355_DEV double floor_(double x) { return std::floor(x); }
M355_DEV double pow_(double x, double y) { return std::pow(x, y); }
M355_DEV double atan_(double x) { return std::atan(x); }
M355_DEV double tan_(double x) { return std::tan(x); }
M355_DEV double pow_l(double x, double lx, double y) { (void)lx; return std::pow(x, y); }
#endif

// k_tab's second map for shapes below 1 (m355_elemwise.h): v = log x (or logit x) then spans hundreds of units -- the
// lower quantiles sit at 1e-9^(1/shape) -- while the function is linear out there and only its bulk needs resolution;
// cells uniform in s = asinh v are ~0.1 wide in v around the bulk and grow like |v| in the tails.
M355_DEV double tab_warp(double v) {
  const double a = __builtin_fabs(v);
  const double s = flog_tab_or_nan(a + sqrt_(__builtin_fma(a, a, 1.0)));
  return v < 0.0 ? -s : s;
}
// c ? tab_warp(v) : v behind a wavefront-uniform branch: with per-lane parameters (k_coltab) the compiler would
// otherwise evaluate the warp for every element, whether or not any lane of the wavefront wants it
M355_DEV double tab_warp_if(bool c, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_amdgcn_ballot_w64(c) == 0) return v;
#endif
  return c ? tab_warp(v) : v;
}
M355_DEV double tab_unwarp(double s) {
  const double e = exp_(s);
  return 0.5 * (e - fdiv(1.0, e));
}

// log Gamma(x + 1) for an observation x (poisson / binomial logpdf: gammaln(x + 1), gammaln(n - x + 1)).  Counts are
// small integers or large: log(x!) from a 128-entry table for integer 0 <= x < 128 (correctly rounded values), Stirling's
// series (DLMF 5.11.1 through z^-11: |error| < 2e-18 at z = x + 1 >= 16) with one logarithm and one reciprocal above,
// the device libm's lgamma for everything else (non-integers below 15, negative arguments and their poles, inf, NaN).
// ~30 fp64-op equivalents instead of ~90: poisson.logpdf goes from lgamma-bound to the HBM stream.
// TAB: the logarithm comes from the LDS table (flog_tab, 14 operations instead of 25) -- for bodies whose kernel has
// filled it (Dist<>::kLogTable; top-level logpdf / pdf only, see k_binned).  The reciprocal of the Stirling correction
// takes one Newton step on the hardware seed (2^-48): the correction is below 1/(12 z) <= 5e-3 where the result is
// above 28, so that is 2e-17 of absolute error.
template <bool TAB = false> M355_DEV double lgamma_obs(double z) {  // log Gamma(z), z = a count + 1 (or a count + a parameter)
  constexpr double kLogFact[128] = {M355_LOGFACT128};
  if ((z >= 1.0) & (z < 129.0)) {
    const int i = (int)z;
    if ((double)i == z) return kLogFact[i - 1];
  }
  if ((z >= 16.0) & (z < 1e300)) {
    double w = hw_rcp(z);
    w = __builtin_fma(__builtin_fma(-z, w, 1.0), w, w);
    const double w2 = w * w;
    double p = -691.0 / 360360.0;
    p = __builtin_fma(p, w2, 1.0 / 1188.0);
    p = __builtin_fma(p, w2, -1.0 / 1680.0);
    p = __builtin_fma(p, w2, 1.0 / 1260.0);
    p = __builtin_fma(p, w2, -1.0 / 360.0);
    p = __builtin_fma(p, w2, 1.0 / 12.0);
    double lz;
    if constexpr (TAB) lz = flog_tab(z);
    else lz = flog_pos(z);
    return (((z - 0.5) * lz - z) + kLogSqrt2Pi) + p * w;
  }
  return lgamma_(z);
}
template <bool TAB = false> M355_DEV double lgamma1p_obs(double x) { return lgamma_obs<TAB>(x + 1.0); }
// log Gamma(z1) + log Gamma(z2) for 129 <= z1, z2 < 1e150 (every lane of the wavefront: the caller votes), table
// logarithms: straight-line code, one reciprocal for both corrections (1/z1 = z2 / (z1 z2)), Stirling through z^-5
// (the next term, 1/(1680 z^7), is below 1e-18 from z = 129).
M355_DEV double lgamma_obs_pair_big(double z1, double z2) {
  double r = hw_rcp(z1 * z2);
  r = __builtin_fma(__builtin_fma(-(z1 * z2), r, 1.0), r, r);
  const double w1 = z2 * r, w2 = z1 * r;
  const double u1 = w1 * w1, u2 = w2 * w2;
  const double c1 = __builtin_fma(__builtin_fma(1.0 / 1260.0, u1, -1.0 / 360.0), u1, 1.0 / 12.0) * w1;
  const double c2 = __builtin_fma(__builtin_fma(1.0 / 1260.0, u2, -1.0 / 360.0), u2, 1.0 / 12.0) * w2;
  const double a1 = __builtin_fma(z1 - 0.5, flog_tab(z1), -z1);
  const double a2 = __builtin_fma(z2 - 0.5, flog_tab(z2), -z2);
  return ((a1 + a2) + 2.0 * kLogSqrt2Pi) + (c1 + c2);
}
// true when every active lane satisfies `c` (host build: the single "lane")
M355_DEV bool wave_all(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __all(c) != 0;
#else
  return c;
#endif
}

// pt.log1mexp: log(1 - exp(x)) for x <= 0.
// ---- erfcx / erfinv, hand-written (tools/gen_erf_coeffs.py; the device libm's versions branch three ways and cost
// ~3x as much: normal.logcdf was 1.05 ms and normal.ppf 0.71 ms per 1e8 elements at lane_util 0.62-0.70) ----------
// erfcx(a) = exp(a^2) erfc(a) for a >= 0, one formula on the whole half line: with t = (a - 4)/(a + 4),
// (1 + a/4) erfcx(a) is a smooth function of t in [-1, 1] (-> 1/sqrt(pi) at a -> inf) and a degree-22 polynomial gives
// it to 4.6e-16; the reciprocal of a + 4 serves both the map and the final scaling.  NaN passes through, erfcx(inf) = 0.
M355_HD_HEAVY double erfcx_huge(double a) { return 0.56418958354775628695 / a; }  // (1/inf = 0, NaN stays NaN)
M355_HD double erfcx_pos(double a) {
  // beyond 2^500 (and for inf / NaN) erfcx(a) = 1 / (a sqrt(pi)) to the last bit: an out-of-line call under a
  // wave-rare branch, so that its division is not evaluated (as a select would) for every lane
#if defined(M355_PROBE_NO_CALL)  // tools/findppf_probe.hip A/B: the rare path inline instead of a call
  if (__builtin_expect(!(a < 0x1p500), 0)) return 0.56418958354775628695 / a;
#else
  if (__builtin_expect(!(a < 0x1p500), 0)) return erfcx_huge(a);
#endif
  const double r = rcp(a + 4.0);
  const double t = (a - 4.0) * r;
  double p = sconst(kErfcxP[kErfcxP_deg]);
#pragma unroll
  for (int k = kErfcxP_deg - 1; k >= 0; --k) p = __builtin_fma(p, t, sconst(kErfcxP[k]));
  return p * (4.0 * r);
}
// erfcx on the whole line (pt.erfcx): erfcx(-a) = 2 exp(a^2) - erfcx(a), with a^2 split exactly so that the
// exponential keeps its relative accuracy for large |x|.
M355_HD double erfcx_(double x) {
  const double a = __builtin_fabs(x);
  const double w = erfcx_pos(a);
  if (!(x < 0.0)) return w;
  const double hi = a * a, lo = __builtin_fma(a, a, -hi);
  const double e = exp_(hi);
  return (e < kInf) ? __builtin_fma(2.0 * e, lo, 2.0 * e) - w : kInf;
}
// erfinv(x) (pt.erfinv), |x| <= 1: x * C(w - 3.125) for w = -log((1 - x)(1 + x)) < 6.25 (|x| < 0.99903: all but one
// lane in a thousand for uniform quantiles), x * T(sqrt(w)) beyond, down to 1 - |x| = 2^-53 (sqrt(w) <= 6.01); both
// polynomials <= 4.2e-16.  |x| = 1 -> +-inf, |x| > 1 -> NaN (the logarithm of a negative number), NaN -> NaN.
M355_DEV double erfinv_(double x) {
  const double w = -flog((1.0 - x) * (1.0 + x));
  double p;
  if (w < 6.25) {
    const double t = w - 3.125;
    p = sconst(kErfinvC[kErfinvC_deg]);
#pragma unroll
    for (int k = kErfinvC_deg - 1; k >= 0; --k) p = __builtin_fma(p, t, sconst(kErfinvC[k]));
  } else {
    const double t = (sqrt_(w) - 4.255) * (1.0 / 1.755);
    p = kErfinvT[kErfinvT_deg];
#pragma unroll
    for (int k = kErfinvT_deg - 1; k >= 0; --k) p = __builtin_fma(p, t, kErfinvT[k]);
    p = (w == kInf) ? kInf : p;
  }
  return p * x;
}

M355_DEV double log1mexp_(double x) {
  return (x < -0.69314718055994530942) ? flog1p(-exp_(x)) : flog(-expm1_(x));
}
// pt.log1pexp (softplus), PyTensor's piecewise form.
M355_DEV double log1pexp_(double x) {
  if (x < -37.0) return exp_(x);
  if (x < 18.0) return flog1p(exp_(x));
  if (x < 33.3) return x + exp_(-x);
  return x;
}
// pt.sigmoid
M355_DEV double sigmoid_(double x) { return 1.0 / (1.0 + exp_(-x)); }

// ------------------------------------------------------------------------------------------------
// log(1+u) - u.  Series form 2*atanh(s) - u with s = u/(2+u) while |s| stays inside the range of
// log_core's polynomial, direct evaluation elsewhere (no cancellation there).
M355_HD double log1pmx(double u) {
  if (u > -0.29 && u < 0.41) {
    const double s = fdiv(u, 2.0 + u);
    const double t = s * s;
    double p = 0x1.2b584aae78c14p-4;
    p = __builtin_fma(p, t, 0x1.39fe606542dbap-4);
    p = __builtin_fma(p, t, 0x1.7462b4ab2ef6cp-4);
    p = __builtin_fma(p, t, 0x1.c71c62e5800a1p-4);
    p = __builtin_fma(p, t, 0x1.2492492df148dp-3);
    p = __builtin_fma(p, t, 0x1.99999999952e2p-3);
    p = __builtin_fma(p, t, 0x1.5555555555558p-2);
    const double R = 2.0 * t * p;  // log(1+u) = 2s + s*R
    const double h = 0.5 * u * u;
    return __builtin_fma(s, h + R, -h);
  }
  return flog1p(u) - u;
}

// Stirling remainder: lgamma(a) - [(a-1/2) log a - a + log(2 pi)/2], a >= 10 (error < 4e-17).
M355_HD double stirling_corr(double a) {
  const double r = rcp(a);
  const double r2 = r * r;
  double s = -3617.0 / 122400.0;
  s = __builtin_fma(s, r2, 1.0 / 156.0);
  s = __builtin_fma(s, r2, -691.0 / 360360.0);
  s = __builtin_fma(s, r2, 1.0 / 1188.0);
  s = __builtin_fma(s, r2, -1.0 / 1680.0);
  s = __builtin_fma(s, r2, 1.0 / 1260.0);
  s = __builtin_fma(s, r2, -1.0 / 360.0);
  s = __builtin_fma(s, r2, 1.0 / 12.0);
  return s * r;
}

// lgamma(1 + a) with relative accuracy for small a > 0 (Taylor: -gamma a + sum zeta(n)(-a)^n/n).
M355_DEV double lgam1p(double a) {
  if (a < 0.03125) {
    double s = -0x1.3b1d971fc5985p-4;
    s = __builtin_fma(s, a, 0x1.556ad63243bc4p-4);
    s = __builtin_fma(s, a, -0x1.748c33114c6d6p-4);
    s = __builtin_fma(s, a, 0x1.9a01e385d5f8fp-4);
    s = __builtin_fma(s, a, -0x1.c806706d57db4p-4);
    s = __builtin_fma(s, a, 0x1.010b36af86397p-3);
    s = __builtin_fma(s, a, -0x1.2703a1dcea3aep-3);
    s = __builtin_fma(s, a, 0x1.5b40cb100c306p-3);
    s = __builtin_fma(s, a, -0x1.a8b9c17aa6149p-3);
    s = __builtin_fma(s, a, 0x1.151322ac7d848p-2);
    s = __builtin_fma(s, a, -0x1.9a4d55beab2d7p-2);
    s = __builtin_fma(s, a, 0x1.a51a6625307d3p-1);
    s = __builtin_fma(s, a, -kEuler);
    return s * a;
  }
  return lgamma_(1.0 + a);
}

// betaln(a,b) = lgamma(a)+lgamma(b)-lgamma(a+b).  When the larger argument is >= 10 the difference
// lgamma(l)-lgamma(s+l) is taken from Stirling's series term by term,
//   -(l-1/2)*(log1p(u)-u) + u/2 - s*log(s+l) + S(l) - S(s+l),  u = s/l,
// so that a small result is not the difference of two huge rounded lgamma values.
M355_HD_HEAVY double lbeta_(double a, double b) {
  const double s = a < b ? a : b, l = a < b ? b : a;
  if (!(l >= 10.0)) return lgamma_(s) + lgamma_(l) - lgamma_(s + l);
  const double u = fdiv(s, l);
  if (s >= 10.0) {
    // both large: Stirling for all three, arranged around log(l / (s + l)) = -log1p(u) and log(s / (s + l)) = log u - log1p(u),
    //   -(l - 1/2) log1p(u) + (s - 1/2) (log u - log1p(u)) - log(s + l)/2 + log(2 pi)/2 + S(s) + S(l) - S(s + l):
    // terms of the size of the result, where lgamma(s) - s log(s + l) are two numbers of the size of s log s (binomial's
    // cdf at n = 1000, p ~ 1/2 carried 1e-12 of relative error from them: round 3, bench_configs.py config 10)
    const double l1p = log1pmx(u) + u;
    return ((-(l - 0.5) * l1p + (s - 0.5) * (flog(u) - l1p)) - 0.5 * flog(s + l)) + kLogSqrt2Pi +
           ((stirling_corr(s) + stirling_corr(l)) - stirling_corr(s + l));
  }
  const double d = -(l - 0.5) * log1pmx(u) + 0.5 * u - s * flog(s + l) + (stirling_corr(l) - stirling_corr(s + l));
  return lgamma_(s) + d;
}

// ------------------------------------------------------------------------------------------------
// log( x^a e^-x / Gamma(a) ), a > 0, 0 < x < inf.  For a >= 10 in Stirling form
//   -a*phi((x-a)/a) + log(a/(2 pi))/2 - S(a),  phi(u) = u - log1p(u),
// which has no cancellation near x ~ a; directly otherwise.  `lga` = lgamma(a).
M355_DEV double log_gamma_kernel(double a, double x, double lga) {
  if (a >= 10.0) {
    const double u = fdiv(x - a, a);
    // log1p(u) - u: series near x ~ a, log(x/a) - u elsewhere (1+u would round x away for x << a)
    const double l = (u > -0.29 && u < 0.41) ? log1pmx(u) : flog(fdiv(x, a)) - u;
    return __builtin_fma(a, l, 0.5 * flog(a * (0.5 / kPi)) - stirling_corr(a));
  }
  return a * flog(x) - x - lga;
}

constexpr int kMaxIter = 100000;

// Iteration counting for the host-side convergence probe (tools/iter_probe.cpp); no-op elsewhere.
#if defined(M355_ITER_PROBE) && !defined(__HIPCC__)
extern long long g_m355_iters, g_m355_outer;
#define M355_ITER() (++g_m355_iters)
#define M355_ITER_OUTER() (++g_m355_outer)
#else
#define M355_ITER() ((void)0)
#define M355_ITER_OUTER() ((void)0)
#endif

// DLMF 8.11.4:  P(a,x) = x^a e^-x / Gamma(a+1) * sum_{n>=0} x^n / ((a+1)...(a+n)).
// Division-free: the partial sums are carried over the common denominator D_n = (a+1)...(a+n),
//   S_n = S_{n-1} (a+n) + x^n,  sum = S_n / D_n,
// so a step is three multiplies / FMAs and the convergence test x^n <= eps S_n needs no quotient (a division costs
// ~10 fp64-op equivalents here, the rest of the step ~5).  S_n >= D_n and S_n >= x^n, so S alone is watched for
// overflow; the rescaling is by an exact power of two (before the test, which is scale-invariant).  One division at the end.
M355_DEV_HEAVY double gammainc_series(double a, double x, double lga) {
  const double lk = log_gamma_kernel(a, x, lga);
  if (lk < kMinLog) return 0.0;
  double N = 1.0, S = 1.0, D = 1.0, r = a;
  for (int n = 0; n < kMaxIter; ++n) {
    M355_ITER();
    r += 1.0;
    N *= x;
    S = __builtin_fma(S, r, N);
    D *= r;
    if (S > 0x1p100) {  // (2^100: the final quotient stays inside the range of the host build's reciprocal seed too)
      N *= 0x1p-100;
      S *= 0x1p-100;
      D *= 0x1p-100;
    }
    if (N <= kEps * S) break;
  }
  return fdiv(S, D) * exp_(lk) / a;
}

// Legendre's continued fraction (DLMF 8.9.2) for Q(a,x), x > a or x > 1: forward recurrence of
// numerators/denominators with rescaling.  The convergence test |p1/q1 - pk/qk| <= eps |pk/qk| is evaluated
// cross-multiplied (no quotient inside the loop); one division at the end.
M355_DEV_HEAVY double gammaincc_cf(double a, double x, double lga) {
  const double lk = log_gamma_kernel(a, x, lga);
  if (lk < kMinLog) return 0.0;
  constexpr double big = 4503599627370496.0, biginv = 2.22044604925031308085e-16;
  double y = 1.0 - a, z = x + y + 1.0, c = 0.0;
  double p2 = 1.0, q2 = x, p1 = x + 1.0, q1 = z * x;
  for (int n = 0; n < kMaxIter; ++n) {
    M355_ITER();
    c += 1.0;
    y += 1.0;
    z += 2.0;
    const double yc = y * c;
    const double pk = p1 * z - p2 * yc;
    const double qk = q1 * z - q2 * yc;
    const double pq = pk * q1;
    const bool done = (qk != 0.0) & (q1 != 0.0) & (__builtin_fabs(__builtin_fma(p1, qk, -pq)) <= kEps * __builtin_fabs(pq));
    p2 = p1;
    p1 = pk;
    q2 = q1;
    q1 = qk;
    if (done) break;
    if (__builtin_fabs(pk) > big) {
      p2 *= biginv;
      p1 *= biginv;
      q2 *= biginv;
      q1 *= biginv;
    }
  }
  return fdiv(p1, q1) * exp_(lk);
}

// Q(a,x) for small x and small a (where 1 - P cancels):
//   Q = [1 - x^a/Gamma(1+a)] - x^a/Gamma(a) * sum_{n>=1} (-x)^n / (n! (a+n)).
M355_DEV_HEAVY double gammaincc_series(double a, double x, double lga) {
  double fac = 1.0, sum = 0.0;
  for (int n = 1; n < kMaxIter; ++n) {
    M355_ITER();
    fac *= -x / (double)n;
    const double term = fdiv(fac, a + (double)n);
    sum += term;
    if (__builtin_fabs(term) <= kEps * __builtin_fabs(sum)) break;
  }
  const double logx = flog(x);
  const double t = -expm1_(a * logx - lgam1p(a));
  return t - exp_(a * logx - lga) * sum;
}

// Temme's uniform asymptotic expansion (DLMF 8.12.7-8; tools/gen_temme_coeffs.py) for a >= kTemmeMinA and
// |x - a| <= kTemmeMaxSigma a -- the transition region, where the series and the continued fraction need O(sqrt(a))
// terms (and more than kMaxIter of them beyond a ~ 1e8) and SciPy's gammainc / gammaincc take this same expansion:
//   eta^2/2 = mu - log(1 + mu), mu = (x - a)/a;   y = a eta^2/2;   T = sum_k c_k(eta) a^-k
//   x <  a:  P = exp(-y) [ erfcx(sqrt y)/2 - T / sqrt(2 pi a) ],  Q = 1 - P
//   x >= a:  Q = exp(-y) [ erfcx(sqrt y)/2 + T / sqrt(2 pi a) ],  P = 1 - Q
// One exponential, one erfcx, ~30-90 FMAs whatever a is; the terms c_k a^-k below 1e-17 are skipped (a >= 800: k <= 5,
// a >= 7e4: k <= 2).  Returns Q when `upper`, else P.
template <int N> M355_HD double temme_poly(const double (&c)[N], double eta) {
  double p = sconst(c[N - 1]);
#pragma unroll
  for (int i = N - 2; i >= 0; --i) p = __builtin_fma(p, eta, sconst(c[i]));
  return p;
}
M355_HD_HEAVY double gammainc_temme(double a, double x, bool upper) {
  const double mu = fdiv(x - a, a);
  const double phi = -log1pmx(mu);
  const double y = a * phi;
  const double eta = (x < a) ? -sqrt_(2.0 * phi) : sqrt_(2.0 * phi);
  const double ia = 1.0 / a;
  double t = 0.0;
  if (a < 800.0) {
    t = temme_poly(kTemmeC8, eta);
    t = __builtin_fma(t, ia, temme_poly(kTemmeC7, eta));
    t = __builtin_fma(t, ia, temme_poly(kTemmeC6, eta));
  }
  if (a < 7e4) {
    t = __builtin_fma(t, ia, temme_poly(kTemmeC5, eta));
    t = __builtin_fma(t, ia, temme_poly(kTemmeC4, eta));
    t = __builtin_fma(t, ia, temme_poly(kTemmeC3, eta));
  }
  t = __builtin_fma(t, ia, temme_poly(kTemmeC2, eta));
  t = __builtin_fma(t, ia, temme_poly(kTemmeC1, eta));
  t = __builtin_fma(t, ia, temme_poly(kTemmeC0, eta));
  const double s = t * sqrt_(ia * (0.5 / kPi));  // T / sqrt(2 pi a)
  const double h = 0.5 * erfcx_pos(sqrt_(y));
  const bool lower_small = x < a;
  const double small = exp_(-y) * (lower_small ? h - s : h + s);
  return (lower_small == upper) ? 1.0 - small : small;
}
M355_HD bool temme_region(double a, double x) {
  return (a >= kTemmeMinA) & (__builtin_fabs(x - a) <= kTemmeMaxSigma * a) & (a < kInf);
}

// Regularised lower incomplete gamma P(a, x) (pt.gammainc).  lga = lgamma(a).
M355_DEV double gammainc_q(double a, double x, double lga);
M355_DEV double gammainc_p(double a, double x, double lga) {
  if (!(x >= 0.0) || !(a >= 0.0)) return kNaN;  // negative or NaN
  if (a == 0.0) return x > 0.0 ? 1.0 : kNaN;
  if (x == 0.0) return 0.0;
  if (a == kInf) return x == kInf ? kNaN : 0.0;
  if (x == kInf) return 1.0;
  if (temme_region(a, x)) return gammainc_temme(a, x, false);
  if (x > 1.0 && x > a) return 1.0 - gammainc_q(a, x, lga);
  return gammainc_series(a, x, lga);
}

// Regularised upper incomplete gamma Q(a, x) (pt.gammaincc).
M355_DEV double gammainc_q(double a, double x, double lga) {
  if (!(x >= 0.0) || !(a >= 0.0)) return kNaN;
  if (a == 0.0) return x > 0.0 ? 0.0 : kNaN;
  if (x == 0.0) return 1.0;
  if (a == kInf) return x == kInf ? kNaN : 1.0;
  if (x == kInf) return 0.0;
  if (temme_region(a, x)) return gammainc_temme(a, x, true);
  if (x > 1.1) {
    return (x < a) ? 1.0 - gammainc_series(a, x, lga) : gammaincc_cf(a, x, lga);
  }
  if (x <= 0.5) {
    return (-0.4 < a * flog(x)) ? gammaincc_series(a, x, lga) : 1.0 - gammainc_series(a, x, lga);
  }
  return (x * 1.1 < a) ? 1.0 - gammainc_series(a, x, lga) : gammaincc_series(a, x, lga);
}

// ------------------------------------------------------------------------------------------------
// Standard normal quantile used only to seed the root finders (|error| < 4.5e-4, A&S 26.2.23).
M355_DEV double norm_quantile_guess(double p) {
  const double pp = p < 0.5 ? p : 1.0 - p;
  const double t = sqrt_(-2.0 * flog(pp));
  const double v = t - fdiv(2.515517 + t * (0.802853 + t * 0.010328),
                            1.0 + t * (1.432788 + t * (0.189269 + t * 0.001308)));
  return p < 0.5 ? -v : v;
}

// Root finders stop after the first step whose relative size is below kRootStep: the iterations
// are at least quadratically convergent (Newton on log-tail, Halley elsewhere), so the error left
// after that step is ~ kRootStep^2 * curvature <= 1e-16.  (Waiting for a step of a few ulp instead
// makes the iterate bounce in the rounding noise of the forward function until the bracket is
// exhausted: measured 6x the work at isolated q, tools/iter_probe.cpp.)
constexpr double kRootStep = 2e-9;
// A Halley step (third order) of relative size s leaves ~K s^3 with K = O(1) in the central region where it is used
// (the tails take the Newton-on-log branches above): 2e-6 leaves ~1e-17 K and saves the evaluation that a 2e-9
// threshold spends on confirming it (gamma.ppf 2.8 -> 2.5, beta.ppf 3.1 -> 2.6 evaluations per quantile).
constexpr double kRootStepHalley = 2e-6;

// Inverse of P(a, .) (pt.gammaincinv): safeguarded Halley iteration on the smaller tail.
// (inlined into its callers like betaincinv_: the binned quantile kernels give it a register budget)
M355_DEV double gammaincinv_p(double a, double p, double lga) {
  if (!(a > 0.0) || !(p >= 0.0) || !(p <= 1.0)) return kNaN;
  if (p == 0.0) return 0.0;
  if (p == 1.0) return kInf;
  const bool upper = p > 0.5;
  const double q = 1.0 - p;  // exact for p in [0.5, 1]
  const double isa = fdiv(1.0, sqrt_(a));
  // The Halley step's cubic error law holds in units of the distribution's width, x / sqrt(a) for a > 1: a step of
  // u widths leaves ~u^3 widths, i.e. u^3 / sqrt(a) of x.  The threshold on the step relative to x scales accordingly
  // (at a = 2e10 a 2e-6 x step is 0.28 widths and would leave 1e-7 x).
  const double halley_tol = a > 1.0 ? kRootStepHalley * isa : kRootStepHalley;
  // Likewise the Newton branches: u widths leave ~u^2 widths = u^2 / sqrt(a) of x, so the step bound goes as a^(-1/4)
  // (measured at a = 2e10 with the fixed 2e-9: roots 1.4e-13 off).
  const double newton_tol = a > 1.0 ? kRootStep * sqrt_(isa) : kRootStep;
  // --- initial guess
  double x;
  {
    const double lg1 = lga + flog(a);  // lgamma(a+1)
    // lower-tail power law  P ~ x^a / Gamma(a+1)
    const double xs = exp_(fdiv(flog(p) + lg1, a));
    if (a < 1.0 || xs < 0.1 * a) {
      x = xs;
      if (upper && a < 1.0) {
        // upper tail for small a: Q ~ x^(a-1) e^-x / Gamma(a)
        const double y = -(flog(q) + lga);
        if (y > 1.0) x = y - (1.0 - a) * flog(y);
      }
    } else {
      // Wilson-Hilferty
      const double z = norm_quantile_guess(p);
      const double w = 1.0 - fdiv(1.0, 9.0 * a) + z * (isa * (1.0 / 3.0));
      x = a * w * w * w;
      if (!(w > 0.0) || x < 1e-3 * a) x = xs;
      if (upper) {
        const double y = -(flog(q) + lga);
        if (y > 3.0 * a + 20.0) x = y + (a - 1.0) * flog(y);
      }
    }
    if (!(x > 0.0)) x = 1e-300;
    if (!(x < kInf)) x = 1e300;
  }
  double lo = 0.0, hi = kInf, shrink = 0.25, grow = 4.0;
  for (int it = 0; it < 200; ++it) {
    M355_ITER_OUTER();
    double f, tail;  // f = P - p or q - Q: increasing in x either way
    if (upper) {
      tail = gammainc_q(a, x, lga);
      f = q - tail;
    } else {
      tail = gammainc_p(a, x, lga);
      f = tail - p;
    }
    if (f == 0.0) break;
    if (f < 0.0) lo = x; else hi = x;
    const double d = exp_(log_gamma_kernel(a, x, lga));  // x * |dP/dx|
    double xn = -1.0;
    bool halley = false;
    if (d > 0.0 && d < kInf) {
      if (!upper && p < 0.25 && tail > 0.0) {
        // Newton on log P(log x): exact for the power-law tail P ~ x^a / Gamma(a+1)
        xn = x * exp_(-fdiv(flog(fdiv(tail, p)) * tail, d));
      } else if (upper && q < 0.25 && tail > 0.0) {
        // Newton on log Q(x): near-exact for the exponential tail
        xn = x + x * fdiv(flog(fdiv(tail, q)) * tail, d);
      } else {
        const double dx_rel = fdiv(f, d);                  // Newton step / x
        const double curv = 0.5 * dx_rel * (a - 1.0 - x);  // Halley: f''/f' = (a-1)/x - 1
        const double den = 1.0 - curv;
        halley = den > 0.5 && den < 2.0;
        xn = x - x * (halley ? fdiv(dx_rel, den) : dx_rel);
      }
    }
    if (xn > 0.0 && __builtin_fabs(xn - x) <= (halley ? halley_tol : newton_tol) * x) {  // converged (also when xn rounds onto a bracket end)
      x = xn;
      break;
    }
    bool safeguarded = false;
    if (!(xn > lo) || !(xn < hi)) {
      safeguarded = true;
      if (hi == kInf) {
        xn = (lo > 0.0 ? lo : x) * grow + 1.0;
        grow *= grow;
      } else if (lo == 0.0) {
        xn = hi * shrink;
        shrink *= shrink;
      } else {
        xn = (hi > 4.0 * lo) ? sqrt_(lo) * sqrt_(hi) : 0.5 * (lo + hi);
      }
      if (!(xn > lo) || !(xn < hi)) break;  // interval exhausted
    }
    const double dx = __builtin_fabs(xn - x);
    x = xn;
    // a small Newton / Halley step squares (or cubes) the error it leaves: converged.  Safeguard
    // (bisection-like) steps only converge linearly and run down to a few ulp.
    if (dx <= (safeguarded ? 4.0 * kEps : (halley ? halley_tol : newton_tol)) * x) break;
  }
  return x;
}

// ------------------------------------------------------------------------------------------------
// log( x^a (1-x)^b / B(a,b) ) with y = 1 - x supplied separately (so that callers that know 1-x
// exactly keep it).  lbeta = lgamma(a)+lgamma(b)-lgamma(a+b).
// Of x and y the smaller one is the accurate one (the callers hold it exactly; the other is 1 - it, rounded), so the
// logarithm of the larger is taken as log1p(-smaller): with log(1 - x) from the rounded 1 - x the term b log y carries
// b * 1.1e-16 of absolute error (5.5e-11 at b = 5e5: measured on betainc(1/2, nu/2, .) at nu = 1e6).
M355_HD double log_beta_kernel(double a, double b, double x, double y, double lbeta) {
  if (a >= 10.0 && b >= 10.0) {
    // both large: around the mode x0 = a / (a + b) the three terms a log x + b log y - lbeta are each of size
    // (a + b) log 2 and cancel to O(1) -- 1e-11 of relative error in I_x at a + b = 1e5 (binomial.cdf at n = 9e4, round 3).
    // With Stirling for B(a, b) and u1 = x / x0 - 1 = (b x - a y) / a, u2 = y / (1 - x0) - 1 = -(b x - a y) / b
    // (a u1 + b u2 = 0 exactly) the kernel is
    //   a (log1p(u1) - u1) + b (log1p(u2) - u2) + log(a b / (2 pi (a + b))) / 2 - S(a) - S(b) + S(a + b),
    // every term of the size of the result.
    const double t = b * x - a * y;
    const double u1 = fdiv(t, a), u2 = fdiv(-t, b);
    const double l1 = (u1 > -0.29 && u1 < 0.41) ? log1pmx(u1) : ((y < 0.5) ? flog1p(-y) : flog(x)) - flog(fdiv(a, a + b)) - u1;
    const double l2 = (u2 > -0.29 && u2 < 0.41) ? log1pmx(u2) : ((x < 0.5) ? flog1p(-x) : flog(y)) - flog(fdiv(b, a + b)) - u2;
    return (a * l1 + b * l2) + 0.5 * flog(fdiv(a * b, (a + b) * (2.0 * kPi))) -
           ((stirling_corr(a) + stirling_corr(b)) - stirling_corr(a + b));
  }
  const double lx = (y < 0.5) ? flog1p(-y) : flog(x);
  const double ly = (x < 0.5) ? flog1p(-x) : flog(y);
  return a * lx + b * ly - lbeta;
}

// Power series for I_x(a,b), good when b*x <= 1 and x <= 0.95 (DLMF 8.17.8 rearranged):
//   I_x = x^a/B(a,b) * [ 1/a + sum_{n>=1} (1-b)(2-b)...(n-b) x^n / (n! (a+n)) ].
M355_HD_HEAVY double betainc_pseries(double a, double b, double x, double lbeta) {
  const double ai = fdiv(1.0, a);
  double t = (1.0 - b) * x;
  double v = fdiv(t, a + 1.0);
  const double t1 = v;
  double s = 0.0, n = 2.0;
  const double z = kEps * ai;
  for (int it = 0; it < kMaxIter && __builtin_fabs(v) > z; ++it) {
    M355_ITER();
    const double u = fdiv((n - b) * x, n);
    t *= u;
    v = fdiv(t, a + n);
    s += v;
    n += 1.0;
  }
  s += t1;
  s += ai;
  const double lt = a * flog(x) - lbeta;
  if (lt > -700.0 && lt < 700.0) return s * exp_(lt);
  const double tl = lt + flog(s);
  return (tl < kMinLog) ? 0.0 : exp_(tl);
}

// Continued fraction DLMF 8.17.22, x < (a+1)/(a+b+2), by the forward recurrence of its numerators and
// denominators A_k = b_k A_{k-1} + a_k A_{k-2} after the equivalence transformation that clears the partial
// numerators' denominators (a_k' = D_{k-1} N_k, b_k' = D_k for a_k = N_k / D_k, b_k = 1): no division inside a step,
// the convergence test is cross-multiplied, the recurrence is rescaled by a power of two when it leaves [2^-100, 2^100],
// one division at the end.  Returns the fraction
// value (without the x^a(1-x)^b/(a B) prefactor).
M355_HD_HEAVY double betainc_cf(double a, double b, double x) {
  const double qab = a + b, qam = a - 1.0;
  // A_{-1} = 1, B_{-1} = 0, A_0 = b_0 = 1, B_0 = 1 for the fraction 1 / (1 + d_1 / (1 + d_2 / (1 + ...))):
  // evaluate f = 1 + d_1/(1 + d_2/(...)) and return 1 / f
  double Am = 1.0, Bm = 0.0, A = 1.0, B = 1.0;  // (A_{k-2}, B_{k-2}), (A_{k-1}, B_{k-1}) of f
  double Dprev = 1.0;
  for (int m = 1; m < kMaxIter; ++m) {
    M355_ITER();
    const double dm = (double)m, m2 = 2.0 * dm;
    // d_{2m-1} = -(a+m-1)(a+b+m-1) x / ((a+2m-2)(a+2m-1)), d_{2m} = m (b-m) x / ((a+2m-1)(a+2m))
    const double N1 = -(qam + dm) * (qab + dm - 1.0) * x, D1 = (qam + m2 - 1.0) * (qam + m2);
    double t = Dprev * N1;
    double An = D1 * A + t * Am, Bn = D1 * B + t * Bm;
    Am = A; Bm = B; A = An; B = Bn;
    const double N2 = dm * (b - dm) * x, D2 = (qam + m2) * (a + m2);
    t = D1 * N2;
    An = D2 * A + t * Am;
    Bn = D2 * B + t * Bm;
    Am = A; Bm = B; A = An; B = Bn;
    Dprev = D2;
    // the D_k grow like (a + 2m)^2: rescale by an exact power of two when the pair leaves [2^-100, 2^100]
    const double mag = __builtin_fabs(A) + __builtin_fabs(B);
    if (mag > 0x1p100 || mag < 0x1p-100) {
      const double sc = mag > 1.0 ? 0x1p-100 : 0x1p100;
      A *= sc; B *= sc; Am *= sc; Bm *= sc;
    }
    // successive convergents Am/Bm and A/B agree to 2 eps: cross-multiplied, no quotient inside the loop
    const double ab = A * Bm;
    if ((B != 0.0) & (Bm != 0.0) & (__builtin_fabs(__builtin_fma(Am, B, -ab)) <= 2.0 * kEps * __builtin_fabs(ab))) break;
  }
  return fdiv(B, A);
}

// I_x(a, b) for a large and b <= 1 with x near 1 (y = 1 - x < 0.3): DiDonato & Morris (ACM TOMS 18 (1992), Algorithm 708,
// BGRAT; the arrangement of Boost's beta_small_b_large_a_series),
//   I_x(a, b) ~ M sum_n p_n J_n(b, u),  u = -T log x,  T = a + (b - 1)/2,  M = H(b, u) Gamma(a + b) / (Gamma(a) T^b),
//   H(b, u) = u^b e^-u / Gamma(b),  J_0 = Q(b, u) / H(b, u),  J_n = (b_n (b_n + 1) J_{n-1} + (u + b_n + 1) (lx/2)^(2n-2)) / 4T^2.
// The leading term is the upper incomplete gamma Q(b, a y): the result keeps its RELATIVE accuracy when I is small, and,
// unlike the continued fraction taken at x -> 1 (whose first partial denominator 1 - x (a + b)/(a + 1) carries the
// rounding of x amplified by 1/y: 3e-17 a of relative error -- Student t's cdf at nu = 1e5 was off by 3e-12, round 3), it
// works from y.  Only where a device or plain host compiler sees it (hipcc's host pass has no gammainc_q; the host
// derives I_{1/2}(a, b) only, which never comes here).
constexpr double kBgratMinA = 100.0;  // below, the continued fraction's 3e-17 a is inside 1e-14
#if (defined(__HIP_DEVICE_COMPILE__) || !defined(__HIPCC__)) && !defined(M355_NO_BGRAT)
#define M355_HAVE_BGRAT 1
M355_DEV_HEAVY double betainc_bgrat(double a, double b, double x, double y) {
  const double bm1 = b - 1.0;
  const double t = a + 0.5 * bm1;
  const double lx = (y < 0.35) ? flog1p(-y) : flog(x);
  const double u = -t * lx;
  const double lgb = lgamma_(b);
  const double lh = b * flog(u) - u - lgb;  // log H(b, u)
  if (!(lh > -700.0)) return 0.0;
  const double h = exp_(lh);
  // Gamma(a + b) / (Gamma(a) T^b) by Stirling: (a - 1/2) log1p(b / a) + b (log(a + b) - log T) - b + S(a + b) - S(a)
  const double lr = (a - 0.5) * flog1p(fdiv(b, a)) + b * flog1p(fdiv(0.5 * (b + 1.0), t)) - b + (stirling_corr(a + b) - stirling_corr(a));
  const double prefix = h * exp_(lr);
  constexpr int kTerms = 12;  // (at most 7 needed over the dispatch region a >= 100, y < 0.3: 2e5 random points)
  double p[kTerms];
  p[0] = 1.0;
  double j = fdiv(gammainc_q(b, u, lgb), h);
  double sum = prefix * j;
  double lx2 = 0.5 * lx;
  lx2 *= lx2;
  double lxp = 1.0;
  const double t4 = 4.0 * t * t;
  double b2n = b;
  double fact_tnp1 = 1.0;  // (2n + 1)!
#pragma unroll 1
  for (int n = 1; n < kTerms; ++n) {
    fact_tnp1 *= (double)(2 * n) * (double)(2 * n + 1);
    double pn = 0.0, f = 6.0;  // f = (2m + 1)!
#pragma unroll 1
    for (int m = 1; m < n; ++m) {
      pn += ((double)m * b - (double)n) * fdiv(p[n - m], f);
      f *= (double)(2 * m + 2) * (double)(2 * m + 3);
    }
    pn = fdiv(pn, (double)n) + fdiv(bm1, fact_tnp1);
    p[n] = pn;
    j = fdiv(b2n * (b2n + 1.0) * j + (u + b2n + 1.0) * lxp, t4);
    lxp *= lx2;
    b2n += 2.0;
    const double r = prefix * pn * j;
    sum += r;
    if (__builtin_fabs(r) <= kEps * __builtin_fabs(sum)) break;
  }
  return sum;
}
#endif

// Regularised incomplete beta I_x(a,b) (pt.betainc).  `y` must be 1-x; lbeta = betaln(a,b).
// ASYM = false: a caller that knows both parameters below kBgratMinA (the table kernels' exact body for Student t,
// nu <= 50) leaves the asymptotic expansion out of its call graph -- it is never taken there, but its registers set the
// budget of the whole kernel (k_tab<studentt> went from five wavefronts per SIMD to four with it).
M355_HD_HEAVY double betainc_xy_body(double a, double b, double x, double y, double lbeta);
template <bool ASYM = true> M355_HD double betainc_xy(double a, double b, double x, double y, double lbeta) {
  if (!(a > 0.0) || !(b > 0.0) || !(x >= 0.0) || !(x <= 1.0)) return kNaN;
  if (x == 0.0) return 0.0;
  if (y == 0.0) return 1.0;  // (x may have rounded to 1 while y is still a tiny positive number)
#ifdef M355_HAVE_BGRAT
  // one parameter large, the other <= 1, the argument at the large parameter's end and past the reach of the power
  // series of the small one (b x <= 1): the asymptotic expansion above (and its mirror image I_x(a, b) = 1 - I_y(b, a)).
  // Decided HERE, in the inlined wrapper, so that the out-of-line body below keeps nothing alive across that call (with the
  // dispatch inside it, the body saved a register pair in AGPRs, which every kernel reaching it then paid for: quantile
  // kernels compiled for three wavefronts per SIMD fell to two).
  if constexpr (ASYM) {
    if (a >= kBgratMinA && b <= 1.0 && y < 0.3 && a * y > 1.0) return betainc_bgrat(a, b, x, y);
    if (b >= kBgratMinA && a <= 1.0 && x < 0.3 && b * x > 1.0) return 1.0 - betainc_bgrat(b, a, y, x);
  }
#endif
  return betainc_xy_body(a, b, x, y, lbeta);
}
M355_HD_HEAVY double betainc_xy_body(double a, double b, double x, double y, double lbeta) {
  // Evaluate on the side where x is below the mean-like threshold (a+1)/(a+b+2), where both
  // expansions converge fast; the power series only while x^n dies quickly (x <= 0.35).
  bool flip = false;
  if (x > fdiv(a + 1.0, a + b + 2.0)) {
    flip = true;
    double t = a; a = b; b = t;
    t = x; x = y; y = t;
  }
  double r;
  if (b * x <= 1.0 && x <= 0.35) {
    r = betainc_pseries(a, b, x, lbeta);
  } else {
    const double lk = log_beta_kernel(a, b, x, y, lbeta);
    r = (lk < kMinLog) ? 0.0 : fdiv(exp_(lk) * betainc_cf(a, b, x), a);
  }
  if (flip) {
    // 1 - r cancels when the answer is small; where the (slow but direct) power series of the
    // unflipped problem applies, use it instead.  After the swap that problem is I_y(b, a).
    if (r > 0.9 && a * y <= 1.0 && y <= 0.95) return betainc_pseries(b, a, y, lbeta);
    r = 1.0 - r;  // (correctly rounded: 1 when the tail is below eps/2, as SciPy's betainc gives)
  }
  return r;
}
template <bool ASYM = true> M355_DEV double betainc_(double a, double b, double x, double lbeta) {
  return betainc_xy<ASYM>(a, b, x, 1.0 - x, lbeta);
}
// I_{1/2}(a, b): which half of [0, 1] the root of I_x(a, b) = p lies in.  Depends on the parameters only, so
// Dist<>::derive computes it once per launch (host) or LDS table entry and hands it to betaincinv_.
M355_HD double betainc_half(double a, double b, double lbeta) { return betainc_xy<false>(a, b, 0.5, 0.5, lbeta); }  // (x = y = 1/2: never the expansion)

// Upper-tail standard normal quantile, Q(z) = pp (0 < pp <= 1/2): the rational guess plus one Newton step on
// erfc (error ~1e-7 z), for seeds that are worth an erfc (the Student-t seed below).
M355_DEV double norm_upper_quantile(double pp) {
  const double z = -norm_quantile_guess(pp);
  const double q = 0.5 * erfc_(z * kInvSqrt2);
  const double phi = exp_(-0.5 * z * z) * 0.39894228040143267794;
  return (phi > 0.0) ? z + fdiv(q - pp, phi) : z;
}

// Initial guess for I_x(a,b) = p, p <= 1/2 (power-law tail, Student-t expansion for b = 1/2 or a = 1/2, or A&S 26.5.22).
M355_DEV double betaincinv_guess(double a, double b, double p, double lbeta) {
  const double xs = exp_(fdiv(flog(p * a) + lbeta, a));  // I ~ x^a/(a B)
  double x = xs;
  if (b == 0.5 || a == 0.5) {
    // I_x(nu/2, 1/2) is the two-sided tail probability of Student's t at t^2 = nu (1 - x) / x (what studentt.ppf /
    // halfstudentt.ppf ask for).  Cornish-Fisher expansion of the t quantile in 1/nu (A&S 26.7.5), accepted while
    // its last term is small: ~1e-5 in the bulk at nu = 5 where the power-law seed is off by 30 %, which takes the
    // Halley iteration from ~4.7 to ~2.7 steps per quantile (tools/iter_probe.cpp).
    const bool t_side = b == 0.5;                     // else I_y(1/2, nu/2) = p: y = 1 - x, tail = 1 - p
    const double nu = 2.0 * (t_side ? a : b);
    const double tail = t_side ? p : 1.0 - p;
    if (tail > 1e-30 && nu >= 1.0) {
      const double z = norm_upper_quantile(0.5 * tail);
      const double z2 = z * z, in = fdiv(1.0, nu);
      const double g1 = z * (z2 + 1.0) * 0.25;
      const double g2 = z * ((5.0 * z2 + 16.0) * z2 + 3.0) * (1.0 / 96.0);
      const double g3 = z * (((3.0 * z2 + 19.0) * z2 + 17.0) * z2 - 15.0) * (1.0 / 384.0);
      const double g4 = z * ((((79.0 * z2 + 776.0) * z2 + 1482.0) * z2 - 1920.0) * z2 - 945.0) * (1.0 / 92160.0);
      const double i2 = in * in;
      const double last = g4 * (i2 * i2);
      const double t = z + g1 * in + g2 * i2 + g3 * (i2 * in) + last;
      if (__builtin_fabs(last) < 0.05 * t) {
        const double t2 = t * t;
        return t_side ? fdiv(nu, nu + t2) : fdiv(t2, nu + t2);
      }
    }
    return xs;
  }
  if (a >= 1.0 && b >= 1.0) {
    const double z = -norm_quantile_guess(p);
    const double lam = (z * z - 3.0) / 6.0;
    const double ta = fdiv(1.0, 2.0 * a - 1.0), tb = fdiv(1.0, 2.0 * b - 1.0);
    const double hh = fdiv(2.0, ta + tb);
    const double w = fdiv(z * sqrt_(hh + lam), hh) - (tb - ta) * (lam + 5.0 / 6.0 - fdiv(2.0, 3.0 * hh));
    x = fdiv(a, a + b * exp_(2.0 * w));
    if (!(x > 0.0) || !(x < 1.0) || xs < 0.05) x = xs;
  }
  return x;
}

// Inverse of I_.(a,b) (pytensor.tensor.math.betaincinv).  The iterate t is x when the root is
// <= 1/2 and y = 1-x otherwise (full relative accuracy of whichever is small).  Safeguarded Newton/Halley, with Newton on log(tail) while the tail is far below 1.
// ihalf = betainc_half(a, b, lbeta) if the caller has it (NaN: evaluated here).
// Inlined into its callers (not __noinline__ like the other heavy functions): a register budget can only be given to a
// kernel, and the binned quantile kernels give one (Dist<>::kMinWaves, m355_elemwise.h).
// y_out (optional) receives 1 - root with the relative accuracy of whichever of the two is small (the iterate itself).
template <bool ASYM = true>
M355_DEV double betaincinv_(double a, double b, double p, double lbeta, double ihalf = kNaN, double* y_out = nullptr) {
  if (y_out) *y_out = kNaN;
  if (!(a > 0.0) || !(b > 0.0) || !(p >= 0.0) || !(p <= 1.0)) return kNaN;
  if (p == 0.0 || p == 1.0 || (a == b && p == 0.5)) {  // (the symmetric median exactly: skew_studentt.ppf(0.5; a = b) must be mu)
    const double r = (p == 0.0) ? 0.0 : (p == 1.0 ? 1.0 : 0.5);
    if (y_out) *y_out = 1.0 - r;
    return r;
  }
  const bool rep_y = p > ((ihalf == ihalf) ? ihalf : betainc_xy<ASYM>(a, b, 0.5, 0.5, lbeta));
  // Equation: I_x(a,b) = p for p <= 1/2, else I_y(b,a) = 1-p (exact), so that the small tail
  // keeps full relative accuracy; betainc_xy takes x and y separately, so either equation can
  // be evaluated from either iterate.
  const bool eq_c = p > 0.5;
  const double target = eq_c ? 1.0 - p : p;
  const bool natural = rep_y == eq_c;  // tail increases with t
  double t;
  {
    // guess from whichever tail is <= 1/2, converted to the iterate's side
    const bool gc = p > 0.5;
    const double g = gc ? betaincinv_guess(b, a, 1.0 - p, lbeta) : betaincinv_guess(a, b, p, lbeta);
    t = (gc == rep_y) ? g : 1.0 - g;
    if (!(t > 0.0)) t = 1e-300;
    if (!(t < 0.5)) t = 0.25;
  }
  double lo = 0.0, hi = 0.5, shrink = 0.25;
  for (int it = 0; it < 200; ++it) {
    M355_ITER_OUTER();
    const double x = rep_y ? 1.0 - t : t;
    const double y = rep_y ? t : 1.0 - t;
    const double tail = eq_c ? betainc_xy<ASYM>(b, a, y, x, lbeta) : betainc_xy<ASYM>(a, b, x, y, lbeta);
    const double g = tail - target;
    if (g == 0.0) break;
    const double f = natural ? g : -g;  // increasing in t
    if (f < 0.0) lo = t; else hi = t;
    const double d = exp_(log_beta_kernel(a, b, x, y, lbeta));  // x (1-x) |dI/dx|
    double tn = -1.0;
    bool halley = false;
    if (d > 0.0 && d < kInf) {
      const double w = fdiv(d, x * y);  // |d tail / dt|
      if (target < 0.25 && tail > 0.0) {
        const double ls = flog(fdiv(tail, target)) * tail;
        tn = natural ? t * exp_(-fdiv(ls, t * w)) : t + fdiv(ls, w);
      } else {
        const double nstep = fdiv(f, w);
        double c2 = fdiv(a - 1.0, x) - fdiv(b - 1.0, y);  // I''/I' in x
        c2 = rep_y ? -c2 : c2;
        const double den = 1.0 - 0.5 * nstep * c2;
        halley = den > 0.5 && den < 2.0;
        tn = t - (halley ? fdiv(nstep, den) : nstep);
      }
    }
    if (tn > 0.0 && __builtin_fabs(tn - t) <= (halley ? kRootStepHalley : kRootStep) * t) {  // converged (also when tn rounds onto a bracket end)
      t = tn;
      break;
    }
    bool safeguarded = false;
    if (!(tn > lo) || !(tn < hi)) {
      safeguarded = true;
      if (lo == 0.0) {
        tn = hi * shrink;
        shrink *= shrink;
      } else {
        tn = (hi > 4.0 * lo) ? sqrt_(lo) * sqrt_(hi) : 0.5 * (lo + hi);
      }
      if (!(tn > lo) || !(tn < hi)) break;
    }
    const double dt = __builtin_fabs(tn - t);
    t = tn;
    if (dt <= (safeguarded ? 4.0 * kEps : (halley ? kRootStepHalley : kRootStep)) * t) break;  // as in gammaincinv_p
  }
  if (y_out) *y_out = rep_y ? t : 1.0 - t;
  return rep_y ? 1.0 - t : t;
}

}  // namespace m355
