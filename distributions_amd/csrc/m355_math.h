// Device math for the elementwise kernels (gfx950).  fp64 VALU runs at half the fp32 rate and the
// bandwidth-bound logpdf kernels have a budget of ~90 fp64 operations per element at 5.6 TB/s, so
// log/log1p are written lean (ocml's correctly-rounded log is ~85 fp64 operations); everything
// here is accurate to a few ulp, far inside the 1e-12 parity bar.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define M355_DEV __device__ __forceinline__
// Host + device: the parameter-only arithmetic of Dist<D>::derive().  For scalar parameters it runs
// once on the host and the constants travel as kernel arguments (SGPRs), so the streaming kernels
// carry no lgamma code and no registers for it.
#define M355_HD __host__ __device__ __forceinline__
// Series / continued fractions / root finders: one out-of-line copy per code object instead of
// one per call site (the call overhead is noise next to their loops).
#define M355_DEV_HEAVY static __device__ __noinline__
#define M355_HD_HEAVY static __host__ __device__ __noinline__
#else
// Host build of the same arithmetic, used only by tests/hostmath (CPU unit tests of the series and
// root finders while no GPU is attached).  Never part of the shipped library.
#include <cmath>
#include <math.h>
#define M355_DEV inline
#define M355_HD inline
#define M355_DEV_HEAVY static inline
#define M355_HD_HEAVY static inline
#endif

namespace m355 {

#if defined(__HIP_DEVICE_COMPILE__)
M355_DEV double hw_rcp(double d) { return __builtin_amdgcn_rcp(d); }
M355_DEV double hw_frexp_mant(double x) { return __builtin_amdgcn_frexp_mant(x); }
M355_DEV int hw_frexp_exp(double x) { return __builtin_amdgcn_frexp_exp(x); }
#else
M355_HD double hw_rcp(double d) { return (double)(1.0f / (float)d); }  // ~24-bit seed like the hardware
M355_HD double hw_frexp_mant(double x) { int e; return (x == x && x - x == 0.0 && x != 0.0) ? ::frexp(x, &e) : x; }
M355_HD int hw_frexp_exp(double x) { int e = 0; if (x == x && x - x == 0.0 && x != 0.0) ::frexp(x, &e); return e; }
#endif

constexpr double kInf = __builtin_huge_val();
constexpr double kNaN = __builtin_nan("");
constexpr double kLn2Hi = 0x1.62e42fefa3800p-1;  // ln2 with 11 trailing zero bits: e*kLn2Hi exact
constexpr double kLn2Lo = 0x1.ef35793c76730p-45;
constexpr double kLogSqrt2Pi = 0.91893853320467274178;
constexpr double kLogPi = 1.14472988584940017414;
constexpr double kSqrt2 = 1.41421356237309504880;
constexpr double kInvSqrt2 = 0.70710678118654752440;

// 1/d to full double precision: hardware seed + two Newton steps.
M355_HD double rcp(double d) {
  double r = hw_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(e, r, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(e, r, r);
  return r;
}

// n/d, ~1 ulp, for finite non-zero d in the normal range (no scaling for overflow/underflow).
M355_HD double fdiv(double n, double d) {
  double r = rcp(d);
  double q = n * r;
  double rem = __builtin_fma(-q, d, n);
  return __builtin_fma(rem, r, q);
}

// log(m) for m in [sqrt(1/2), sqrt(2)) as 2*atanh(s), s = (m-1)/(m+1); returns (2s, s*t*g(t)*2)
// split so that callers add the small part first.  Coefficients: tools/gen_log_coeffs.py
// (degree 6, max relative error 4.6e-18).
M355_HD void log_core(double m, double& hi, double& lo) {
  const double f = m - 1.0;
  const double s = fdiv(f, m + 1.0);
  const double t = s * s;
  double p = 0x1.2b584aae78c14p-4;
  p = __builtin_fma(p, t, 0x1.39fe606542dbap-4);
  p = __builtin_fma(p, t, 0x1.7462b4ab2ef6cp-4);
  p = __builtin_fma(p, t, 0x1.c71c62e5800a1p-4);
  p = __builtin_fma(p, t, 0x1.2492492df148dp-3);
  p = __builtin_fma(p, t, 0x1.99999999952e2p-3);
  p = __builtin_fma(p, t, 0x1.5555555555558p-2);
  hi = 2.0 * s;
  lo = hi * t * p;
}

// Natural log, all IEEE special cases (x<0 -> NaN, 0 -> -inf, +inf -> +inf, NaN -> NaN,
// subnormals handled by v_frexp).
M355_HD double flog(double x) {
  double m = hw_frexp_mant(x);  // [0.5, 1)
  int e = hw_frexp_exp(x);
  const bool low = m < kInvSqrt2;
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  double hi, lo;
  log_core(m, hi, lo);
  const double ef = (double)e;
  double r = __builtin_fma(ef, kLn2Lo, lo);
  r = r + hi;
  r = __builtin_fma(ef, kLn2Hi, r);
  r = (x < 0.0) ? kNaN : r;
  r = (x == 0.0) ? -kInf : r;
  r = (x == kInf) ? kInf : r;  // NaN input: already NaN through the arithmetic
  return r;
}

// log(x) for callers that guarantee 0 < x < inf (normal or subnormal): no special-case selects.
M355_HD double flog_pos(double x) {
  double m = hw_frexp_mant(x);
  int e = hw_frexp_exp(x);
  const bool low = m < kInvSqrt2;
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  double hi, lo;
  log_core(m, hi, lo);
  const double ef = (double)e;
  double r = __builtin_fma(ef, kLn2Lo, lo);
  r = r + hi;
  return __builtin_fma(ef, kLn2Hi, r);
}

// Keeps a polynomial coefficient in a scalar register pair: without this the compiler
// rematerialises every 64-bit literal with two v_mov_b32 per use (7 fp64-op equivalents per log).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(M355_NO_SCONST)  // (M355_NO_SCONST: tools/findppf_probe.hip's A/B build)
M355_DEV double sconst(double v) {
  asm("" : "+s"(v));
  return v;
}
#else
M355_HD double sconst(double v) { return v; }
#endif

// n/d with NR Newton steps on the hardware seed and an optional residual correction.
template <int NR, bool CORR> M355_HD double fdiv_t(double n, double d) {
  double r = hw_rcp(d);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
  }
  double q = n * r;
  if (CORR) q = __builtin_fma(__builtin_fma(-q, d, n), r, q);
  return q;
}

// Streaming log: exponent/mantissa split with 32-bit integer ops on the high word (half the issue
// cost of fp64 instructions), one wave-rare branch to flog() for zero / subnormal / negative /
// inf / NaN inputs.  Same polynomial as log_core.
template <int NR, bool CORR> M355_HD double flog_fast_t(double x) {
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, x);
  unsigned hx = (unsigned)(bits >> 32);
  if ((hx - 0x00100000u) >= 0x7fe00000u) return flog(x);
  hx += 0x3ff00000u - 0x3fe6a09eu;
  const int k = (int)(hx >> 20) - 0x3ff;
  hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
  const double m = __builtin_bit_cast(double, ((unsigned long long)hx << 32) | (bits & 0xffffffffull));
  const double s = fdiv_t<NR, CORR>(m - 1.0, m + 1.0);
  const double t = s * s;
  double p = sconst(0x1.2b584aae78c14p-4);
  p = __builtin_fma(p, t, sconst(0x1.39fe606542dbap-4));
  p = __builtin_fma(p, t, sconst(0x1.7462b4ab2ef6cp-4));
  p = __builtin_fma(p, t, sconst(0x1.c71c62e5800a1p-4));
  p = __builtin_fma(p, t, sconst(0x1.2492492df148dp-3));
  p = __builtin_fma(p, t, sconst(0x1.99999999952e2p-3));
  p = __builtin_fma(p, t, sconst(0x1.5555555555558p-2));
  const double hi = s + s;
  const double ef = (double)k;
  double r = __builtin_fma(ef, sconst(kLn2Lo), hi * t * p);
  r = r + hi;
  return __builtin_fma(ef, sconst(kLn2Hi), r);
}

// The log used by the bandwidth-bound logpdf bodies: one Newton step on the 24-bit v_rcp_f64 seed
// (measured: 2.1e-15 relative after one step, tools/probe.hip), no residual correction.
M355_HD double flog_stream(double x) { return flog_fast_t<1, false>(x); }

// log(1 + w) for w >= 0 on the streaming path: log(u) + c/u with u = fl(1+w), c its rounding
// error and 1/u ~ 2-u (exact to w^2, applied only where w < 1/2; beyond that c/u < 0.3 ulp of
// the result).
M355_HD double flog1p_stream(double w) {
  const double u = 1.0 + w;
  const double c = w - (u - 1.0);
  const double corr = (w < 0.5) ? c * (2.0 - u) : 0.0;
  return flog_stream(u) + corr;
}

// log(1 + w) with relative accuracy for small |w| (w >= -1; w < -1 -> NaN).
M355_HD double flog1p(double w) {
  const double u = 1.0 + w;
  const double c = w - (u - 1.0);  // rounding error of u (exact for |w| < 2^52)
  const double l = flog(u);
  const bool ok = (u > 0.0) & (u < kInf);
  const double corr = ok ? c * hw_rcp(u) : 0.0;
  return l + corr;
}

// Table log for the bandwidth-bound logpdf bodies with two logarithms per element (beta) or little VALU slack:
// the mantissa interval [sqrt(1/2), sqrt(2)) is cut into 256 cells by the top 8 bits of the (re-based) high word;
// log(m) = log(c_i) + log1p(r), r = m / c_i - 1 by one fma with the tabulated 1 / c_i, |r| <= 2e-3, degree-5
// series.  No division: ~14 fp64-op equivalents + one ds_read_b128 instead of ~25.  Absolute error <= 3e-16
// (the cell that contains 1 has c_i = 1, so r = m - 1 is exact there).  The table (256 x {1/c, log c} = 4 KiB)
// lives in LDS; a kernel whose Dist declares kLogTable calls log_table_init() before the first evaluation.
constexpr int kLogCells = 256;
M355_HD void log_table_entry(int i, double& inv_c, double& log_c) {
  // cell i covers high words [base + i * 4096, base + (i + 1) * 4096), base = 0x3fe6a09e
  const unsigned lo_hw = 0x3fe6a09eu + (unsigned)i * 4096u, hi_hw = lo_hw + 4096u;
  const double lo = __builtin_bit_cast(double, (unsigned long long)lo_hw << 32);
  const double hi = __builtin_bit_cast(double, (unsigned long long)hi_hw << 32);
  const bool has_one = (lo <= 1.0) & (hi > 1.0);
  const double c = has_one ? 1.0 : 0.5 * (lo + hi);
  inv_c = has_one ? 1.0 : 1.0 / c;
  // log(c) is paired with the ROUNDED reciprocal: log(m) = -log(inv_c) + log1p(m * inv_c - 1) exactly
  log_c = has_one ? 0.0 : -flog(inv_c);
}
#if defined(__HIP_DEVICE_COMPILE__)
static __shared__ double g_log_table[2 * kLogCells];
M355_DEV void log_table_init() {
  for (int i = threadIdx.x; i < kLogCells; i += blockDim.x) log_table_entry(i, g_log_table[2 * i], g_log_table[2 * i + 1]);
  __syncthreads();
}
M355_DEV void log_table_fetch(unsigned cell, double& inv_c, double& log_c) {
  inv_c = g_log_table[2 * cell];
  log_c = g_log_table[2 * cell + 1];
}
#else
M355_HD void log_table_init() {}
M355_HD void log_table_fetch(unsigned cell, double& inv_c, double& log_c) { log_table_entry((int)cell, inv_c, log_c); }
#endif
M355_HD double flog_tab(double x) {
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, x);
  unsigned hx = (unsigned)(bits >> 32);
  if ((hx - 0x00100000u) >= 0x7fe00000u) return flog(x);
  hx += 0x3ff00000u - 0x3fe6a09eu;
  const int k = (int)(hx >> 20) - 0x3ff;
  const unsigned frac = hx & 0x000fffffu;
  const double m = __builtin_bit_cast(double, ((unsigned long long)(frac + 0x3fe6a09eu) << 32) | (bits & 0xffffffffull));
  double inv_c, log_c;
  log_table_fetch(frac >> 12, inv_c, log_c);
  const double r = __builtin_fma(m, inv_c, -1.0);
  double p = sconst(0.2);
  p = __builtin_fma(-p, r, sconst(0.25));
  p = __builtin_fma(-p, r, sconst(0x1.5555555555555p-2));
  p = __builtin_fma(-p, r, sconst(0.5));
  const double l1p = __builtin_fma(-(r * r), p, r);  // r - r^2 (1/2 - r (1/3 - r (1/4 - r/5)))
  const double ef = (double)k;
  double res = __builtin_fma(ef, sconst(kLn2Lo), l1p);
  res = res + log_c;
  return __builtin_fma(ef, sconst(kLn2Hi), res);
}
// k_dtab (m355_elemwise.h): the integer window a discrete family proposes for its lookup table
constexpr int kDTabMax = 2048;
// window around (mean, sd) clipped to [slo, shi]; false if it does not fit
M355_HD bool dtab_window(double mean, double sd, double slo, double shi, double& lo, int& len) {
  if (!(mean - mean == 0.0) || !(sd - sd == 0.0)) return false;
  double a = floor(mean - 9.0 * sd - 12.0), b = ceil(mean + 14.0 * sd + 40.0);
  a = a < slo ? slo : a;
  b = b > shi ? shi : b;
  if (!(b >= a) || b - a + 1.0 > (double)kDTabMax) return false;
  lo = a;
  len = (int)(b - a + 1.0);
  return true;
}

// 2^-e with 2^e ~ m (m > 0 finite and normal; 1 otherwise): an exact scale for k_tab's log variables
M355_HD double pow2_near_inv(double m) {
  const int e = (m > 0x1p-900 && m < 0x1p900) ? hw_frexp_exp(m) : 0;
  return __builtin_bit_cast(double, (unsigned long long)(1023 - e) << 52);
}
// flog_tab without its branch: NaN for every argument flog_tab hands to flog (x <= 0, subnormal, inf, NaN).  For
// callers that treat those as misses anyway (k_tab's table variable).
M355_HD double flog_tab_or_nan(double x) {
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, x);
  unsigned hx = (unsigned)(bits >> 32);
  const bool exc = (hx - 0x00100000u) >= 0x7fe00000u;
  hx += 0x3ff00000u - 0x3fe6a09eu;
  const int k = (int)(hx >> 20) - 0x3ff;
  const unsigned frac = hx & 0x000fffffu;
  const double m = __builtin_bit_cast(double, ((unsigned long long)(frac + 0x3fe6a09eu) << 32) | (bits & 0xffffffffull));
  double inv_c, log_c;
  log_table_fetch(frac >> 12, inv_c, log_c);
  const double r = __builtin_fma(m, inv_c, -1.0);
  double p = sconst(0.2);
  p = __builtin_fma(-p, r, sconst(0.25));
  p = __builtin_fma(-p, r, sconst(0x1.5555555555555p-2));
  p = __builtin_fma(-p, r, sconst(0.5));
  const double l1p = __builtin_fma(-(r * r), p, r);
  const double ef = (double)k;
  double res = __builtin_fma(ef, sconst(kLn2Lo), l1p);
  res = res + log_c;
  res = __builtin_fma(ef, sconst(kLn2Hi), res);
  return exc ? kNaN : res;
}
// log(1 + w), w > -1, on the table log (same correction as flog1p_stream; for w < 0 the sum 1 + w is exact
// below -1/2 and the first-order correction c (2 - u) ~ c / u holds to 3e-17 above it)
M355_HD double flog1p_tab(double w) {
  const double u = 1.0 + w;
  const double c = w - (u - 1.0);
  const double corr = (w < 0.5) ? c * (2.0 - u) : 0.0;
  return flog_tab(u) + corr;
}

// Sum of logarithms carried as a product: sum_i log(v_i) = log(prod_i m_i) + ln2 * sum_i e_i with
// v_i = m_i 2^e_i, m_i in [sqrt(1/2), sqrt(2)) (the split of flog_tab: v ~ 1 gives e = 0, so nothing cancels
// between the two parts).  One fp64 multiply and five integer ops per term instead of a logarithm
// (~14-25 op equivalents); the logarithm is taken once per thread.  Every multiply rounds at 2^-53 relative,
// i.e. an absolute error per term of the size of one rounded logarithm of a value of order 1 - callers use it
// only where the coefficient of the logarithm is moderate (see Dist<>::SumAcc).  Terms must be positive,
// finite and normal (everything else goes through the elementwise function); renorm() at least every
// 1024 terms keeps the running product inside the exponent range.
struct LogProd {
  double m = 1.0;  // product of the re-based mantissas
  int e = 0;       // sum of the exponents
  M355_HD void mul(double v) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
    const unsigned hx = (unsigned)(bits >> 32) + (0x3ff00000u - 0x3fe6a09eu);
    e += (int)(hx >> 20) - 0x3ff;
    m *= __builtin_bit_cast(double, ((unsigned long long)((hx & 0x000fffffu) + 0x3fe6a09eu) << 32) | (bits & 0xffffffffull));
  }
  M355_HD void renorm() {
    const double t = m;
    m = 1.0;
    mul(t);
  }
  M355_HD double log() const {
    LogProd c = *this;
    c.renorm();
    const double ef = (double)c.e;
    return __builtin_fma(ef, kLn2Hi, __builtin_fma(ef, kLn2Lo, flog(c.m)));
  }
};
// the smallest normal double: below it LogProd::mul does not apply
constexpr double kMinNormal = 0x1p-1022;
// largest |coefficient of the logarithm| for which the sum-logp kernels use LogProd: above it the absolute
// rounding of the product (coef * 1e-16 per term) is not negligible next to terms of order 1
constexpr double kSumAccMaxCoef = 1024.0;

// pytensor.tensor.xlogx.xlogy0 given log(y): 0 where a == 0.
M355_HD double xlogy0_l(double a, double logy) { return a == 0.0 ? 0.0 : a * logy; }

}  // namespace m355
