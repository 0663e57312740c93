// Per-distribution scalar bodies: what one element of the reference's fused Elemwise loop
// computes.  Dist<D>::derive() turns the raw parameters into the loop-invariant constants
// (gammaln/log of parameters, reciprocals) and Dist<D>::eval<FN>() is the per-element body.
// pt.switch is a select: only the selected arm's value is produced (identical result, the dead
// arm's NaNs never leak).  file:line citations refer to /root/reference/pytensor_distributions.
#pragma once
#include "m355_math.h"
#include "m355_logphitab.h"
#include "m355_special.h"
#include "../../include/m355_dist.h"

namespace m355 {

template <int D> struct Dist;

// k_tab / k_coltab (m355_elemwise.h) look cdf / sf up through the table of the LOG function the reference composes them
// from -- e.g. gamma.py:40-41 cdf = exp(logcdf), gamma.py:53-54 sf = 1 - cdf, beta.py:82-83 sf = exp(logsf) -- so that a
// small probability keeps the relative accuracy of its logarithm (a table of the probability itself is validated at
// 1e-14 absolute only).  Dist<>::tab_route<FN>() names the tabulated function and the epilogue, FN = M355_CDF / M355_SF.
enum { kTabEpiNone = 0, kTabEpiExp = 1, kTabEpiOneMinusExp = 2 };
struct TabRoute {
  int fn, epi;
};

// ------------------------------------------------------------------------------- helper.py:29-54
M355_DEV double ppf_bounds_cont(double v, double q, double lower, double upper) {
  double r = (q == 1.0) ? upper : v;
  r = (q == 0.0) ? lower : r;
  return ((q < 0.0) | (q > 1.0)) ? kNaN : r;
}
// helper.py:4-26
M355_DEV double cdf_bounds(double p, double x, double lower, double upper) {
  double r = (x >= upper) ? 1.0 : p;
  return (x < lower) ? 0.0 : r;
}
// helper.py:90-91
M355_DEV double sf_bounds(double v, double q, double lower, double upper) {
  double r = (q > upper) ? 0.0 : v;
  return (q < lower) ? 1.0 : r;
}

// --------------------------------------------------------------------------------------- normal
// normal.logcdf body on a standardised z (normal.py:70-76):
//   z < -1: log(erfcx(-z/sqrt2)/2) - z^2/2,   else: log1p(-erfc(z/sqrt2)/2).
// One evaluation of w = erfcx(|z|/sqrt2) serves every z, without a branch: for z < 0 (both reference arms: on
// -1 <= z < 0 the second arm is log1p(-(2 - erfc(|z|/sqrt2))/2) = log(w/2) - z^2/2, the same number to rounding)
// the result is log(w/2) - z^2/2; for z >= 0 it is log1p(u), u = -w exp(-z^2/2)/2, taken as log(1 + u) plus the
// rounding of 1 + u -- so a single logarithm, of a per-lane selected argument, covers both signs.
M355_DEV double normal_logcdf_z(double z) {
  const double h = 0.5 * z * z;
  const double w = erfcx_pos(__builtin_fabs(z) * kInvSqrt2);
  const bool neg = z < 0.0;
  const double u = -0.5 * w * exp_(-h);
  const double f1 = 1.0 + u;
  const double f = neg ? 0.5 * w : f1;
  double c = neg ? 0.0 : u - (f1 - 1.0);
  c = ((f > 0.0) & (f < kInf)) ? c * hw_rcp(f) : 0.0;
  const double l = flog_fast_t<1, true>(f) + c;  // one Newton step + residual correction: the quotient is exact to 2^-96
  return neg ? l - h : l;
}
// (x - mu) / sigma as the reference divides, from the host's correctly rounded 1/sigma: product plus one residual
// correction (<= 1 ulp from the IEEE quotient, three operations instead of the ~14 of v_div_scale / fmas / fixup).
M355_DEV double standardise(double d, double sigma, double inv_sigma) {
  const double q = d * inv_sigma;
  const double v = __builtin_fma(__builtin_fma(-q, sigma, d), inv_sigma, q);
  return (q - q == 0.0) ? v : q;  // inf / NaN quotients pass through (the residual would be inf - inf)
}

// log Phi(z) from a piecewise polynomial (tools/gen_logphi_table.py -> m355_logphitab.h: 256 cells on [-10, 6), degree
// 8, 2.2e-16 of max(1, |log Phi|)) held in LDS: ~8 operations for the cell and nine fused multiply-adds instead of an
// erfcx, an exponential and a logarithm (~160).  For the kernels with scalar parameters (k_scalar fills the table);
// z outside the table, NaN included, takes normal_logcdf_z.
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const double g_phi_table_rom[kPhiCells * kPhiStride] = {M355_LOGPHI_TABLE};
static __shared__ double g_phi_table[kPhiCells * kPhiStride];
M355_DEV void phi_table_init() {
  for (int i = threadIdx.x; i < kPhiCells * kPhiStride; i += blockDim.x) g_phi_table[i] = g_phi_table_rom[i];
  __syncthreads();
}
M355_DEV const double* phi_cell(int i) { return g_phi_table + i * kPhiStride; }
#else
static const double g_phi_table_rom[kPhiCells * kPhiStride] = {M355_LOGPHI_TABLE};
M355_HD void phi_table_init() {}
M355_HD const double* phi_cell(int i) { return g_phi_table_rom + i * kPhiStride; }
#endif
M355_DEV double normal_logcdf_z_tab(double z) {
  const double s = __builtin_fma(z, kPhiCellsPerUnit, -kPhiLo * kPhiCellsPerUnit);  // cell number + position in it
  if (!((s >= 0.0) & (s < (double)kPhiCells))) return normal_logcdf_z(z);
  const double fi = __builtin_floor(s);
  const double t = __builtin_fma(2.0, s - fi, -1.0);
  const double* c = phi_cell((int)fi);
  double p = c[kPhiDeg];
#pragma unroll
  for (int k = kPhiDeg - 1; k >= 0; --k) p = __builtin_fma(p, t, c[k]);
  return p;
}

template <> struct Dist<M355_NORMAL> {
  static constexpr int NP = 2;
  // k_scalar evaluates these through eval_tab after phi_table_init()
  static constexpr bool tab_fn(int fn) { return fn == M355_LOGCDF || fn == M355_LOGSF || fn == M355_SF; }

  struct K {
    double mu, sigma, inv_sigma, c;
  };
  static M355_HD K derive(const double* p) {
    K k;
    k.mu = p[0];
    k.sigma = p[1];
    k.inv_sigma = 1.0 / p[1];
    k.c = -kLogSqrt2Pi - flog(p[1]);
    return k;
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) {  // normal.py:79-80
      const double z = (x - k.mu) * k.inv_sigma;
      return __builtin_fma(-0.5 * z, z, k.c);
    } else if constexpr (FN == M355_PDF) {  // normal.py:54-55
      const double z = (x - k.mu) / k.sigma;
      return 1.0 / sqrt_(2.0 * kPi * k.sigma * k.sigma) * exp_(-0.5 * z * z);
    } else if constexpr (FN == M355_LOGCDF) {  // normal.py:70-76
      return normal_logcdf_z(standardise(x - k.mu, k.sigma, k.inv_sigma));
    } else if constexpr (FN == M355_LOGSF) {  // normal.py:83-84: logcdf(-x, -mu, sigma)
      return normal_logcdf_z(standardise(-x + k.mu, k.sigma, k.inv_sigma));
    } else if constexpr (FN == M355_SF) {  // normal.py:62-63
      return exp_(normal_logcdf_z((-x + k.mu) / k.sigma));
    } else if constexpr (FN == M355_CDF) {  // normal.py:46-47
      return 0.5 * (1.0 + erf_((x - k.mu) / (k.sigma * kSqrt2)));
    } else if constexpr (FN == M355_PPF || FN == M355_ISF) {  // normal.py:58-59, 50-51
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      const double v = k.mu + k.sigma * kSqrt2 * erfinv_(2.0 * q - 1.0);
      return ppf_bounds_cont(v, q, -kInf, kInf);
    }
    return kNaN;
  }
  template <int FN> static M355_DEV double eval_tab(double x, const K& k) {
    if constexpr (FN == M355_LOGCDF) return normal_logcdf_z_tab(standardise(x - k.mu, k.sigma, k.inv_sigma));
    else if constexpr (FN == M355_LOGSF) return normal_logcdf_z_tab(standardise(-x + k.mu, k.sigma, k.inv_sigma));
    else return exp_(normal_logcdf_z_tab((-x + k.mu) / k.sigma));  // M355_SF
  }
};

// ---------------------------------------------------------------------------------------- gamma
template <> struct Dist<M355_GAMMA> {
  static constexpr int kMinWaves = 3;  // binned kernel: 168-VGPR budget around the inlined gammaincinv_p (m355_elemwise.h)
  static constexpr int kBinned = 1;  // logcdf / logsf / ppf ...: lanes diverge, see k_binned (m355_elemwise.h)
  static constexpr int NP = 2;
  static constexpr bool kLogTable = true;  // logpdf uses flog_tab
  struct K {
    double alpha, beta, am1, c, lga, tsc;  // tsc: 2^-e with 2^e ~ alpha / beta (k_tab's variable is log(x tsc))
  };
  static M355_HD K derive(const double* p) {
    K k;
    k.alpha = p[0];
    k.beta = p[1];
    k.am1 = p[0] - 1.0;
    k.lga = lgamma_(p[0]);
    k.c = p[0] * flog(p[1]) - k.lga;
    const double mean = p[0] / p[1];
    const int e = (mean > 0x1p-900 && mean < 0x1p900) ? hw_frexp_exp(mean) : 0;
    k.tsc = __builtin_bit_cast(double, (unsigned long long)(1023 - e) << 52);
    return k;
  }
  // k_tab (m355_elemwise.h): the cdf family is tabulated in w = log x.
  // Very large shapes neither: the exact bodies' own noise (~1e-16 alpha) then exceeds what a cell must show to be used.
  static M355_HD bool tab_ok(const K& k) { return k.alpha >= 0.01 && k.alpha <= 1e4 && k.beta > 0.0 && k.beta < kInf; }
  // (quantiles: ppf ~ q^(1/alpha) at the lower end, which no polynomial in logit q follows for alpha < 1: log(ppf) is
  // tabulated instead, tab_ppf_log)
  static constexpr bool kTabOkFn = true;
  template <int FN> static M355_HD bool tab_ok_fn(const K& k) { return (FN != M355_PPF && FN != M355_ISF) || k.alpha >= 0.05; }
  static M355_HD bool tab_ppf_log(const K& k) { return k.alpha < 1.0; }  // (below 0.05, ppf(1e-9) underflows)
  // gamma.py:40-41 cdf = exp(logcdf); gamma.py:53-54 sf = 1 - cdf
  template <int FN> static constexpr TabRoute tab_route() {
    return FN == M355_CDF ? TabRoute{M355_LOGCDF, kTabEpiExp} : TabRoute{M355_LOGCDF, kTabEpiOneMinusExp};
  }
  // (x is scaled by a power of two near beta / alpha first, so that |w| stays of the order of the spread of log x: for
  // large alpha the function is steep in w -- d log P / dw ~ sqrt(alpha) -- and log x itself would carry |log x| ulps;
  // alpha < 1: warped, see tab_warp)
  static M355_DEV double tab_w(double x, const K& k) {
    const double v = flog_tab_or_nan(x * k.tsc);
    return tab_warp_if(k.alpha < 1.0, v);
  }
  static M355_DEV double tab_x(double w, const K& k) { return fdiv(exp_(k.alpha < 1.0 ? tab_unwarp(w) : w), k.tsc); }
  // gamma.py:77-82 with (alpha-1)*log(x) as xlogy0 (quirk Q2).
  static M355_DEV double logpdf(double x, const K& k) {
    const double t = xlogy0_l(k.am1, flog_tab(x));
    const double r = (t - x * k.beta) + k.c;
    return (x < 0.0) ? -kInf : r;
  }
  // sum of logpdf over a thread's elements (k_scalar with REDUCE): sum_i log x_i as a LogProd.
  // Same value as summing logpdf() within the rounding of the sum itself; x <= 0, subnormal, inf and NaN
  // take logpdf() and add into `exc`, which carries -inf / NaN exactly as the plain sum would.
  struct SumAcc {
    LogProd px;
    double sx = 0.0, exc = 0.0;
    int cnt = 0;
  };
  static M355_DEV bool sum_ok(const K& k) { return __builtin_fabs(k.am1) <= kSumAccMaxCoef; }
  static M355_DEV void sum_add(SumAcc& s, double x, const K& k) {
    if ((x >= kMinNormal) & (x < kInf)) {
      s.px.mul(x);
      s.sx += x;
      ++s.cnt;
    } else {
      s.exc += logpdf(x, k);
    }
  }
  static M355_DEV void sum_renorm(SumAcc& s) { s.px.renorm(); }
  static M355_DEV double sum_finish(const SumAcc& s, const K& k) {
    const double reg = (k.am1 * s.px.log() - k.beta * s.sx) + (double)s.cnt * k.c;
    return reg + s.exc;
  }
  // gamma.py:65-74
  static M355_DEV double logcdf(double x, const K& k) {
    const double xb = x * k.beta;
    double r;
    if (xb < k.alpha) {
      r = flog(gammainc_p(k.alpha, xb, k.lga));
    } else {
      r = flog1p(-gammainc_q(k.alpha, xb, k.lga));
    }
    return (x < 0.0) ? -kInf : r;
  }
  // gamma.py:85-94
  static M355_DEV double logsf(double x, const K& k) {
    const double xb = x * k.beta;
    double r;
    if (xb < k.alpha) {
      r = flog1p(-gammainc_p(k.alpha, xb, k.lga));
    } else {
      r = flog(gammainc_q(k.alpha, xb, k.lga));
    }
    return (x < 0.0) ? 0.0 : r;
  }
  // gamma.py:48-50
  static M355_DEV double ppf(double q, const K& k) {
    return ppf_bounds_cont(gammaincinv_p(k.alpha, q, k.lga) / k.beta, q, 0.0, kInf);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));        // gamma.py:44-45
    if constexpr (FN == M355_LOGCDF) return logcdf(x, k);
    if constexpr (FN == M355_LOGSF) return logsf(x, k);
    if constexpr (FN == M355_CDF) return exp_(logcdf(x, k));        // gamma.py:40-41
    if constexpr (FN == M355_SF) return 1.0 - exp_(logcdf(x, k));   // gamma.py:53-54
    if constexpr (FN == M355_PPF) return ppf(x, k);
    if constexpr (FN == M355_ISF) return ppf(1.0 - x, k);           // gamma.py:57-58
    return kNaN;
  }
};

// ----------------------------------------------------------------------------------------- beta
template <> struct Dist<M355_BETA> {
  static constexpr int kMinWaves = 3;  // binned kernel: 168-VGPR budget around the inlined betaincinv_ (m355_elemwise.h)
  static constexpr int kBinned = 1;  // logcdf / logsf / ppf ...: lanes diverge, see k_binned (m355_elemwise.h)
  static constexpr int NP = 2;
  static constexpr bool kLogTable = true;  // logpdf uses flog_tab
  struct K {
    double a, b, am1, bm1, lbeta, ihalf;  // ihalf = I_{1/2}(a, b), for betaincinv_
  };
  static M355_HD K derive(const double* p) {
    K k;
    k.a = p[0];
    k.b = p[1];
    k.am1 = p[0] - 1.0;
    k.bm1 = p[1] - 1.0;
    // pytensor.tensor.special.betaln (lgamma(a)+lgamma(b)-lgamma(a+b)), cancellation-free form
    k.lbeta = lbeta_(p[0], p[1]);
    k.ihalf = betainc_half(p[0], p[1], k.lbeta);
    return k;
  }
  // beta.py:102-108; xlogy0(alpha+beta-1, 1) is a*log(1) == 0 (or the xlogy0 zero).
  // k_tab: the cdf family is tabulated in w = logit x (1 - x is exact for x >= 1/2); shapes below 1 as for gamma
  static M355_HD bool tab_ok(const K& k) { return k.a >= 0.01 && k.b >= 0.01 && k.a + k.b <= 500.0; }
  // (quantiles as for gamma; logsf / sf with alpha < 1/2: the reference's betainc at the rounded 1 - x is a staircase
  // below x ~ 1e-6, where such a shape puts > 0.1 % of the data -- the cells there fail validation by design, and that
  // many misses cost more than the table saves)
  static constexpr bool kTabOkFn = true;
  template <int FN> static M355_HD bool tab_ok_fn(const K& k) {
    // (b < 1: the upper nodes' Newton steps on logsf are ill-conditioned at x -> 1.  a >= 0.1: round 3's sweep of the box
    // measured 4.8e-13 relative between table and exact kernel at (0.066, 12.05) and (0.080, 156.2) -- log(ppf) has slope
    // 1/a in logit q, and a cell's dropped coefficients are judged against sum |c|, which grows with that slope)
    if (FN == M355_PPF || FN == M355_ISF) return k.a >= 0.1 && k.b >= 1.0;
    if (FN == M355_LOGSF || FN == M355_SF) return k.a >= 0.5;
    return true;
  }
  // betainc's own rounding noise grows with the shapes (the prefactor x^a y^b / B carries ~1e-16 (a + b)): 9e-14 at
  // (150, 300) against the node function itself (tests/hostmath hm_tab_build2) -- with four dropped coefficients as the
  // measure (round 3) 69 % of that table's cells failed 1e-14.  2.5e-16 (a + b) from a + b = 40: 1.1e-13 at 450, which
  // keeps |table - exact kernel| below 3e-13 and both inside the 1e-12 bar.
  // (only with both shapes >= 1: below, logsf follows the reference's staircase -- betainc at the rounded 1 - x -- whose
  // cells must go on failing at 1e-14; at (0.5, 499.5) the looser tolerance let a cell pass that sat 6e-13 off it)
  static M355_HD double tab_tol(const K& k) {
    return 1e-14 * ((k.a >= 1.0 && k.b >= 1.0 && k.a + k.b > 40.0) ? (k.a + k.b) * (1.0 / 40.0) : 1.0);
  }
  static M355_HD bool tab_warped(const K& k) { return (k.a < 1.0) | (k.b < 1.0); }
  // beta.py:66-67 cdf = exp(logcdf); beta.py:82-83 sf = exp(logsf)
  template <int FN> static constexpr TabRoute tab_route() {
    return FN == M355_CDF ? TabRoute{M355_LOGCDF, kTabEpiExp} : TabRoute{M355_LOGSF, kTabEpiExp};
  }
  // (range ends when even ppf(1e-3) / ppf(1 - 1e-3) round to 0 / 1: the outermost doubles inside the support)
  static M355_HD bool tab_ppf_log(const K& k) { return k.a < 1.0; }
  static M355_HD double tab_cap(bool upper, const K&) { return upper ? 1.0 - 0x1p-53 : 0x1p-1000; }
  static M355_DEV double tab_w(double x, const K& k) {
    const double v = flog_tab_or_nan(fdiv_t<1, true>(x, 1.0 - x));
    return tab_warp_if(tab_warped(k), v);
  }
  static M355_DEV double tab_x(double w, const K& k) {
    // the smaller of x, 1 - x from exp(-|v|) (for |v| in the hundreds 1 / (1 + exp(-v)) would round to 1)
    const double v = tab_warped(k) ? tab_unwarp(w) : w;
    const double e = exp_(-__builtin_fabs(v));
    const double sm = fdiv(e, 1.0 + e);
    return v < 0.0 ? sm : 1.0 - sm;
  }
  // Node values.  Below 1/2 the argument x = sm is a double to full relative accuracy and the node is the exact body
  // (where the reference's own logsf is a staircase -- betainc at the rounded 1 - x, x < 1e-10 -- the cell then fails
  // validation and those elements keep the exact body).  Above 1/2 the node's x = 1 - sm would be rounded to the grid
  // of doubles next to 1, which the DATA is too but a node of the smooth interpolant must not be: betainc takes x and
  // its complement separately, so the node passes sm as the complement (for a double x the reference's 1 - x is exact
  // there, so this is the function the reference computes).
  static constexpr bool kTabNode = true;
  template <int FN> static M355_DEV double tab_node(double w, const K& k) {
    const double v = tab_warped(k) ? tab_unwarp(w) : w;
    const double e = exp_(-__builtin_fabs(v));
    const double sm = fdiv(e, 1.0 + e);
    if (v < 0.0) return eval<FN, false>(sm, k);
    constexpr bool UP = FN == M355_LOGSF || FN == M355_SF;
    // (<false>: inside tab_ok, a + b <= 500, the continued fraction's 3e-17 a is within 1.5e-14 -- see tab_miss)
    const double I = UP ? betainc_xy<false>(k.b, k.a, sm, 1.0 - sm, k.lbeta) : betainc_xy<false>(k.a, k.b, 1.0 - sm, sm, k.lbeta);
    return (FN == M355_LOGCDF || FN == M355_LOGSF) ? flog(I) : I;
  }
  static M355_DEV double logpdf(double x, const K& k) {
    const double t = xlogy0_l(k.am1, flog_tab(x)) + xlogy0_l(k.bm1, flog_tab(1.0 - x));
    const double r = t - (0.0 + k.lbeta);
    return ((x < 0.0) | (x > 1.0)) ? -kInf : r;
  }
  // float32 data (k_scalar_f32, tolerance 1e-5): the two logarithms in float32 (v_log_f32 through ocml's log2f: 1 ulp,
  // denormal inputs scaled), the combination in float64.  The error is 1.2e-7 (|am1 ln x| + |bm1 ln(1 - x)|); next to the
  // mode those terms cancel against betaln, so the form is used only while the coefficients are small enough that
  // the cancelled magnitude stays below ~30 (f32_ok: |alpha - 1|, |beta - 1| <= 8, |betaln| <= 12): error <= 4e-6
  // absolute in the worst cell (tests/test_gpu_elemwise.py::test_float32_arithmetic_bodies sweeps the parameter box).
  // Two float64 table logarithms cost 405 cycles per wavefront and element, this ~130: the float32 stream becomes
  // HBM-bound (3.1 -> 6 TB/s at 8 B / element).
  static M355_HD bool f32_ok(const K& k) {
    return (__builtin_fabs(k.am1) <= 8.0) & (__builtin_fabs(k.bm1) <= 8.0) & (__builtin_fabs(k.lbeta) <= 12.0);
  }
  static M355_DEV float logpdf_f32(float x, const K& k) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double lx = (double)::log2f(x), lc = (double)::log2f(1.0f - x);
#else
    const double lx = (double)std::log2((float)x), lc = (double)std::log2(1.0f - x);
#endif
    constexpr double kLn2 = 0.69314718055994530942;
    const double ta = (k.am1 == 0.0) ? 0.0 : (k.am1 * kLn2) * lx;  // xlogy0
    const double tb = (k.bm1 == 0.0) ? 0.0 : (k.bm1 * kLn2) * lc;
    const double r = (ta + tb) - k.lbeta;
    return ((x < 0.0f) | (x > 1.0f)) ? -__builtin_huge_valf() : (float)r;
  }
  // sum of logpdf (see Dist<M355_GAMMA>::SumAcc): sum log x and sum log(1 - x) as LogProds; for
  // min_normal <= x < 1 the complement 1 - x is in [2^-53, 1), normal.
  struct SumAcc {
    LogProd px, pc;
    double exc = 0.0;
    int cnt = 0;
  };
  static M355_DEV bool sum_ok(const K& k) {
    return (__builtin_fabs(k.am1) <= kSumAccMaxCoef) & (__builtin_fabs(k.bm1) <= kSumAccMaxCoef);
  }
  static M355_DEV void sum_add(SumAcc& s, double x, const K& k) {
    if ((x >= kMinNormal) & (x < 1.0)) {
      s.px.mul(x);
      s.pc.mul(1.0 - x);
      ++s.cnt;
    } else {
      s.exc += logpdf(x, k);
    }
  }
  static M355_DEV void sum_renorm(SumAcc& s) {
    s.px.renorm();
    s.pc.renorm();
  }
  static M355_DEV double sum_finish(const SumAcc& s, const K& k) {
    const double reg = (k.am1 * s.px.log() + k.bm1 * s.pc.log()) - (double)s.cnt * k.lbeta;
    return reg + s.exc;
  }
  // beta.py:90-99
  template <bool ASYM = true> static M355_DEV double logcdf(double x, const K& k) {
    if (x < 0.0) return -kInf;
    if (x < 1.0) return flog(betainc_<ASYM>(k.a, k.b, x, k.lbeta));
    return (x == x) ? 0.0 : 0.0;  // x >= 1 and NaN (all comparisons false) both select 0
  }
  // beta.py:111-120
  template <bool ASYM = true> static M355_DEV double logsf(double x, const K& k) {
    if (x < 0.0) return 0.0;
    if (x < 1.0) return flog(betainc_<ASYM>(k.b, k.a, 1.0 - x, k.lbeta));
    return -kInf;
  }
  template <int FN, bool ASYM = true> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));  // beta.py:74-75
    if constexpr (FN == M355_LOGCDF) return logcdf<ASYM>(x, k);
    if constexpr (FN == M355_LOGSF) return logsf<ASYM>(x, k);
    if constexpr (FN == M355_CDF) return exp_(logcdf<ASYM>(x, k));  // beta.py:66-67
    if constexpr (FN == M355_SF) return exp_(logsf<ASYM>(x, k));    // beta.py:82-83
    if constexpr (FN == M355_PPF || FN == M355_ISF) {         // beta.py:78-79, 70-71
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_bounds_cont(betaincinv_<ASYM>(k.a, k.b, q, k.lbeta, k.ihalf), q, 0.0, 1.0);
    }
    return kNaN;
  }
  // the exact body inside tab_ok (a + b <= 500): without betainc's asymptotic expansion for a parameter >= 100, whose
  // registers and scratch would otherwise set the table kernels' budget (it improves 1.5e-14 to 1e-15 there: not needed)
  template <int FN> static M355_DEV double tab_miss(double x, const K& k) { return eval<FN, false>(x, k); }
};

// ------------------------------------------------------------------------------------- studentt
template <> struct Dist<M355_STUDENTT> {
  static constexpr int kMinWaves = 3;  // binned kernel: 168-VGPR budget around the inlined betaincinv_ (m355_elemwise.h)
  static constexpr int kBinned = 1;  // logcdf / logsf / ppf ...: lanes diverge, see k_binned (m355_elemwise.h)
  static constexpr int NP = 3;
  static constexpr bool kLogTable = true;  // logpdf uses flog1p_tab
  struct K {
    double nu, mu, sigma, c, hnp1, inv_s2nu, lbeta_h, ihalf_h;  // lbeta_h = betaln(nu/2, 1/2), ihalf_h = I_{1/2}(nu/2, 1/2)
  };
  static M355_HD K derive(const double* p) {
    K k;
    k.nu = p[0];
    k.mu = p[1];
    k.sigma = p[2];
    const double lg_h = lgamma_(0.5 * p[0]);
    const double lg_hp = lgamma_(0.5 * (p[0] + 1.0));
    k.c = lg_hp - lg_h - 0.5 * flog(p[0] * kPi * p[2] * p[2]);
    k.hnp1 = 0.5 * (p[0] + 1.0);
    k.inv_s2nu = 1.0 / (p[2] * p[2] * p[0]);
    k.lbeta_h = lbeta_(0.5 * p[0], 0.5);
    k.ihalf_h = betainc_half(0.5 * p[0], 0.5, k.lbeta_h);
    return k;
  }
  // ---- k_tab (m355_elemwise.h) ----
  // The cdf family is tabulated in w = asinh z (the tails are linear in w), the quantiles in logit q.  The reference's
  // own formulas lose digits next to the centre -- betainc at nu / (z^2 + nu), which rounds (studentt.py:100-105), and
  // sqrt(1/b - 1) with b next to 1 (studentt.py:79-89) -- and the exact bodies above reproduce them as they are, which
  // makes them staircases there: unusable as interpolation nodes.  The nodes therefore come from the same functions
  // taken on their accurate side (tab_node / tab_ppf_node: betainc_xy with z^2 / (z^2 + nu) supplied, the complement of
  // the root from betaincinv_), and for the cdf family the band around the centre where the reference's rounding
  // exceeds a fifth of the parity bar -- |z| < 3e-4 sqrt(nu) / B(1/2, nu/2), 2e-4 of the mass at nu = 5 -- is a miss:
  // those elements take the exact body, staircase included.  (Quantiles need no band: their parity is judged with the
  // modelled rounding of the reference's sqrt(1/b - 1), tests/harness.py _student_window.)
  // (nu <= 50: the band grows like nu -- 3 % of the mass at nu = 300 -- and the quantile's nodes get noisy with it)
  static M355_HD bool tab_ok(const K& k) { return k.nu >= 1.0 && k.nu <= 50.0 && k.sigma > 0.0 && k.sigma < kInf && k.mu - k.mu == 0.0; }
  // studentt.py:67-68 cdf = exp(logcdf); studentt.py:92-93 sf = exp(logsf)
  template <int FN> static constexpr TabRoute tab_route() {
    return FN == M355_CDF ? TabRoute{M355_LOGCDF, kTabEpiExp} : TabRoute{M355_LOGSF, kTabEpiExp};
  }
  static M355_DEV double tab_w(double x, const K& k) {
    const double z = fdiv_t<1, true>(x - k.mu, k.sigma), a = __builtin_fabs(z);
    const double w = flog_tab_or_nan(a + sqrt_(__builtin_fma(a, a, 1.0)));
    const double band = 3e-4 * sqrt_(k.nu) * exp_(-k.lbeta_h);  // (uniform: hoisted out of the loops)
    return (a < band) ? kNaN : (z < 0.0 ? -w : w);
  }
  static M355_DEV double tab_x(double w, const K& k) {
    const double e = exp_(w);
    return __builtin_fma(k.sigma, 0.5 * (e - fdiv(1.0, e)), k.mu);
  }
  // cdf family at a node, smooth: I_x(nu/2, 1/2) with x = nu / (z^2 + nu) and y = z^2 / (z^2 + nu) both formed directly
  static constexpr bool kTabNode = true;
  template <int FN> static M355_DEV double tab_node(double w, const K& k) {
    const double e = exp_(w);
    double z = 0.5 * (e - fdiv(1.0, e));
    if (FN == M355_LOGSF || FN == M355_SF) z = -z;
    const double d = __builtin_fma(z, z, k.nu);
    const double f = 0.5 * betainc_xy<false>(0.5 * k.nu, 0.5, fdiv(k.nu, d), fdiv(z * z, d), k.lbeta_h);  // (nu <= 50: tab_miss)
    const double l = (z < 0.0) ? flog(f) : flog1p(-f);
    return (FN == M355_CDF || FN == M355_SF) ? exp_(l) : l;
  }
  // quantiles within 1e-6 of the median are misses: the reference returns exactly mu for |q - 1/2| < 1e-14 and values
  // quantised in steps of ~1e-8 sigma next to it (b rounds to 1 - k 2^-53), which the exact body follows
  static M355_DEV bool tab_q_miss(double q) { return __builtin_fabs(q - 0.5) < 1e-6; }
  // quantile at a node, smooth: p = sigmoid(-|w|) is the smaller tail; b = the root of I_b(nu/2, 1/2) = 2p with its
  // complement y, |z| = sqrt(nu y / b)
  static M355_DEV double tab_ppf_node(double w, const K& k) {
    const double e = exp_(-__builtin_fabs(w));
    const double p = fdiv(e, 1.0 + e);
    double y;
    const double b = betaincinv_<false>(0.5 * k.nu, 0.5, 2.0 * p, k.lbeta_h, k.ihalf_h, &y);
    const double r = sqrt_(k.nu) * sqrt_(fdiv(y, b));
    return __builtin_fma(w < 0.0 ? -r : r, k.sigma, k.mu);
  }
  // studentt.py:108-119.  The nu > 1e5 arm is normal.LOGCDF(x, mu, sigma): the reference binds
  // `normal_logpdf` to normal.logcdf (studentt.py:8); reproduced as is (quirk Q1).
  static M355_DEV double logpdf(double x, const K& k) {
    if (k.nu > 1e5) return normal_logcdf_z((x - k.mu) / k.sigma);
    const double d = x - k.mu;
    return k.c - k.hnp1 * flog1p_tab(d * d * k.inv_s2nu);
  }
  // sum of logpdf (see Dist<M355_GAMMA>::SumAcc): sum log1p(w_i) = log prod (1 + w_i) + the rounding of
  // 1 + w_i recovered to first order (as flog1p_tab does), summed on the side.
  struct SumAcc {
    LogProd pu;
    double corr = 0.0, exc = 0.0;
    int cnt = 0;
  };
  static M355_DEV bool sum_ok(const K& k) { return (k.hnp1 <= kSumAccMaxCoef) & !(k.nu > 1e5); }
  static M355_DEV void sum_add(SumAcc& s, double x, const K& k) {
    const double d = x - k.mu;
    const double w = d * d * k.inv_s2nu;
    if (w < 0x1p1000) {
      const double u = 1.0 + w;
      const double c = w - (u - 1.0);
      s.corr += (w < 0.5) ? c * (2.0 - u) : 0.0;
      s.pu.mul(u);
      ++s.cnt;
    } else {
      s.exc += logpdf(x, k);
    }
  }
  static M355_DEV void sum_renorm(SumAcc& s) { s.pu.renorm(); }
  static M355_DEV double sum_finish(const SumAcc& s, const K& k) {
    const double reg = (double)s.cnt * k.c - k.hnp1 * (s.pu.log() + s.corr);
    return reg + s.exc;
  }
  // studentt.py:100-105.  The nu > 1e5 arm is normal.CDF(z, mu, sigma) on the already
  // standardised z (mis-import studentt.py:6); reproduced as is (quirk Q1).
  template <bool ASYM = true> static M355_DEV double logcdf_raw(double x, double mu, const K& k) {
    const double z = (x - mu) / k.sigma;
    if (k.nu > 1e5) return 0.5 * (1.0 + erf_((z - mu) / (k.sigma * kSqrt2)));
    const double f = 0.5 * betainc_<ASYM>(0.5 * k.nu, 0.5, k.nu / (z * z + k.nu), k.lbeta_h);
    return (z < 0.0) ? flog(f) : flog1p(-f);
  }
  // the exact body behind a table miss (m355_elemwise.h tab_exact): tab_ok has nu <= 50, so betainc's asymptotic
  // expansion for a parameter >= 100 stays out of the table kernels' call graph (same values)
  template <int FN> static M355_DEV double tab_miss(double x, const K& k) {
    if constexpr (FN == M355_LOGCDF) return logcdf_raw<false>(x, k.mu, k);
    if constexpr (FN == M355_LOGSF) return logcdf_raw<false>(-x, -k.mu, k);
    if constexpr (FN == M355_CDF) return exp_(logcdf_raw<false>(x, k.mu, k));
    if constexpr (FN == M355_SF) return exp_(logcdf_raw<false>(-x, -k.mu, k));
    if constexpr (FN == M355_PPF) return ppf<false>(x, k);
    if constexpr (FN == M355_ISF) return ppf<false>(1.0 - x, k);
    return eval<FN>(x, k);
  }
  template <bool ASYM = true> static M355_DEV double ppf(double q, const K& k) {  // studentt.py:79-89
    double r;
    if (__builtin_fabs(q - 0.5) < 1e-14) {
      r = 0.0;
    } else {
      const double qq = (q < 0.5) ? 2.0 * q : 2.0 * (1.0 - q);
      const double b = betaincinv_<ASYM>(0.5 * k.nu, 0.5, qq, k.lbeta_h, k.ihalf_h);
      const double t = sqrt_(k.nu) * sqrt_(1.0 / b - 1.0);
      r = (q < 0.5) ? -t : t;
    }
    return ppf_bounds_cont(k.mu + r * k.sigma, q, -kInf, kInf);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));               // studentt.py:75-76
    if constexpr (FN == M355_LOGCDF) return logcdf_raw(x, k.mu, k);
    if constexpr (FN == M355_LOGSF) return logcdf_raw(-x, -k.mu, k);       // studentt.py:122-123
    if constexpr (FN == M355_CDF) return exp_(logcdf_raw(x, k.mu, k));     // studentt.py:67-68
    if constexpr (FN == M355_SF) return exp_(logcdf_raw(-x, -k.mu, k));    // studentt.py:92-93
    if constexpr (FN == M355_PPF) return ppf(x, k);
    if constexpr (FN == M355_ISF) return ppf(1.0 - x, k);                  // studentt.py:71-72
    return kNaN;
  }
};

// helper.py:57-87
M355_DEV double ppf_bounds_disc(double v, double q, double lower, double upper) {
  double r = (q == 1.0) ? upper : v;
  r = (q == 0.0) ? lower - 1.0 : r;
  return ((q < 0.0) | (q > 1.0)) ? kNaN : r;
}

// find_ppf_discrete (optimization.py:102-129): integer Newton / unit-step search from floor(x0),
// max_iter = 100, tol = 1e-7.  The reference iterates the whole batch until every element is
// `found`; a found element is a fixed point of the step (x_new = x_int), so a per-lane loop with
// early exit gives the same values.  Only q in (0, 1) reaches the loop result: every other q is
// decided by ppf_bounds_disc (a NaN q turns the iterate into NaN in the first step).
template <class CDF, class PMF>
M355_DEV double find_ppf_discrete(double q, double x0, double lower, double upper, CDF cdf, PMF pmf) {
  double x = floor_(x0);
  if (!(q == q)) return kNaN;
  if ((q > 0.0) & (q < 1.0)) {
    for (int it = 0; it < 100; ++it) {
      const double xi = floor_(x);
      const double c = cdf(xi), cm = cdf(xi - 1.0);
      x = xi;
      if ((c >= q - 1e-7) & (cm < q)) break;  // found
      const double pm = pmf(xi);
      const double delta = (c - q) / (pm > 1e-10 ? pm : 1e-10);
      const double sgn = (c > q) ? 1.0 : ((c < q) ? -1.0 : 0.0);
      const double step = (__builtin_fabs(delta) < 0.5) ? sgn : floor_(delta);
      const double xn = xi - step;
      x = xn < lower ? lower : (xn > upper ? upper : xn);  // pt.clip
    }
  }
  return ppf_bounds_disc(x, q, lower, upper);
}

// -------------------------------------------------------------------------------------- poisson
template <> struct Dist<M355_POISSON> {
  static constexpr int kBinned = 1;
  static constexpr int NP = 1;
  struct K {
    double mu, logmu;
  };
  static M355_HD K derive(const double* p) {
    K k;
    k.mu = p[0];
    k.logmu = flog(p[0]);
    return k;
  }
  // poisson.py:70-71: no support switch; gammaln's poles give -inf at negative integers (Q6).
  static M355_DEV double logpdf(double x, const K& k) {
    return (xlogy0_l(x, k.logmu) - lgamma1p_obs(x)) - k.mu;
  }
  // k_dtab (m355_elemwise.h): window of integers for the cdf family
  static M355_HD bool dtab_range(const K& k, double& lo, int& len) { return dtab_window(k.mu, sqrt(k.mu), 0.0, kInf, lo, len); }
  static M355_DEV double cdf(double x, const K& k) {  // poisson.py:49-50
    return cdf_bounds(gammainc_q(x + 1.0, k.mu, lgamma_(x + 1.0)), x, 0.0, kInf);
  }
  static M355_DEV double sf(double x, const K& k) {  // poisson.py:58-59
    const double a = floor_(x) + 1.0;
    return sf_bounds(gammainc_p(a, k.mu, lgamma_(a)), x, 0.0, kInf);
  }
  // the pieces of the quantile search, named so that k_dtab can run the same search on tabulated values
  static M355_DEV double dtab_cdf(double v, const K& k) { return cdf(v, k); }
  static M355_DEV double dtab_pmf(double v, const K& k) { return exp_(logpdf(v, k)); }
  template <class C, class P> static M355_DEV double ppf_with(double q, const K& k, C c, P p) {
    return find_ppf_discrete(q, k.mu, 0.0, kInf, c, p);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf(x, k));  // poisson.py:45-46
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return sf(x, k);
    if constexpr (FN == M355_LOGCDF) return flog(cdf(x, k));  // poisson.py:74-75
    if constexpr (FN == M355_LOGSF) return flog(sf(x, k));    // poisson.py:78-79
    if constexpr (FN == M355_PPF || FN == M355_ISF) {         // poisson.py:53-55, 62-63
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_with(q, k, [&](double v) { return dtab_cdf(v, k); }, [&](double v) { return dtab_pmf(v, k); });
    }
    return kNaN;
  }
};

// ------------------------------------------------------------------------------------- binomial
template <> struct Dist<M355_BINOMIAL> {
  static constexpr int kBinned = 1;
  static constexpr int NP = 2;
  struct K {
    double n, p, lgn1, logp, log1mp;
  };
  static M355_HD K derive(const double* p) {
    K k;
    k.n = p[0];
    k.p = p[1];
    k.lgn1 = lgamma_(p[0] + 1.0);
    k.logp = flog(p[1]);
    k.log1mp = flog1p(-p[1]);
    return k;
  }
  static constexpr bool kLogTable = true;  // (top-level logpdf / pdf only: the pmf inside ppf runs where k_binned has no table)
  template <bool TAB> static M355_DEV double logpdf_t(double x, const K& k) {  // binomial.py:66-75
    const double nmx = k.n - x;
    if constexpr (TAB) {
      // counts and their complements both large (a wave-uniform fact for most data): the two log-factorials together
      if (wave_all((x >= 128.0) & (nmx >= 128.0) & (k.n < 1e150))) {
        const double r = (k.lgn1 - lgamma_obs_pair_big(x + 1.0, nmx + 1.0)) + __builtin_fma(x, k.logp, nmx * k.log1mp);
        return r;  // (0 < x < n here)
      }
    }
    double r = k.lgn1 - lgamma1p_obs<TAB>(x) - lgamma1p_obs<TAB>(nmx) + xlogy0_l(x, k.logp) + nmx * k.log1mp;
    return ((x < 0.0) | (x > k.n)) ? -kInf : r;
  }
  static M355_DEV double logpdf(double x, const K& k) { return logpdf_t<false>(x, k); }
  static M355_HD bool dtab_range(const K& k, double& lo, int& len) {
    return dtab_window(k.n * k.p, sqrt(k.n * k.p * (1.0 - k.p)), 0.0, k.n, lo, len);
  }
  static M355_DEV double cdf(double x, const K& k) {  // binomial.py:45-46
    const double a = k.n - x, b = x + 1.0;
    return cdf_bounds(betainc_(a, b, 1.0 - k.p, lbeta_(a, b)), x, 0.0, k.n);
  }
  // the pieces of the quantile search, named so that k_dtab can run the same search on tabulated values
  static M355_DEV double dtab_cdf(double v, const K& k) { return cdf(v, k); }
  static M355_DEV double dtab_pmf(double v, const K& k) { return exp_(logpdf(v, k)); }
  template <class C, class P> static M355_DEV double ppf_with(double q, const K& k, C c, P p) {
    return find_ppf_discrete(q, k.n * k.p, 0.0, k.n, c, p);
  }
  template <int FN> static M355_DEV double eval(double x, const K& k) {
    if constexpr (FN == M355_LOGPDF) return logpdf_t<true>(x, k);
    if constexpr (FN == M355_PDF) return exp_(logpdf_t<true>(x, k));    // binomial.py:41-42
    if constexpr (FN == M355_CDF) return cdf(x, k);
    if constexpr (FN == M355_SF) return 1.0 - cdf(x, k);        // binomial.py:54-55
    if constexpr (FN == M355_LOGCDF) return flog(cdf(x, k));    // binomial.py:78-79
    if constexpr (FN == M355_LOGSF) return flog1p(-cdf(x, k));  // binomial.py:82-83
    if constexpr (FN == M355_PPF || FN == M355_ISF) {           // binomial.py:49-51, 58-59 (x0 = mean = n p)
      const double q = (FN == M355_ISF) ? 1.0 - x : x;
      return ppf_with(q, k, [&](double v) { return dtab_cdf(v, k); }, [&](double v) { return dtab_pmf(v, k); });
    }
    return kNaN;
  }
};

// Which (dist, fn) pairs exist on the path: all eight functions of all six distributions.
constexpr bool supported(int dist, int fn) {
  if (fn < 0 || fn > M355_ISF) return false;
  return dist >= M355_NORMAL && dist < M355_DIST_COUNT;
}

}  // namespace m355
