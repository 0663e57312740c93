// Counter-based sampling (the replacement for numpy.random.Generator behind pt.random.X:
// normal.py:66-67, gamma.py:61-62, beta.py:86-87, studentt.py:96-97, poisson.py:66-67,
// binomial.py:62-63).  Philox4x32-10 (Salmon et al., SC'11) keyed by the 64-bit seed; the counter
// is (global element index lo, hi, attempt, stream id), so a draw is a pure function of
// (seed, stream, element index, parameters): the concatenated output is identical for any
// partition of the index range over GPUs, and any range can be regenerated (resume) from its
// offset.  Samplers: ziggurat normal (Box-Muller inside the composite samplers), Marsaglia-Tsang gamma (alpha < 1 by the U^(1/alpha)
// boost), beta as a gamma ratio, Student t as normal / sqrt(chi2/nu), Poisson by sequential
// inversion (mu < 10) or PTRS (Hoermann 1993), binomial by sequential inversion (n*min(p,q) < 10)
// or BTRS (Hoermann 1993).  Rejection loops run per lane with a bounded attempt count.
#pragma once
#include "m355_dists4.h"
#include "m355_sincoscoef.h"
#ifdef M355_ZIGTAB_HEADER  // tools/zig_probe.hip: another layer count
#include M355_ZIGTAB_HEADER
#else
#include "m355_zigtab.h"
#endif

namespace m355 {

struct Philox {
  unsigned c[4];
};

M355_HD unsigned mulhi32(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
// a ^ b ^ c in one instruction (gfx950's v_bitop3_b32, truth table 0x96): half of Philox's forty xors
M355_HD unsigned xor3(unsigned a, unsigned b, unsigned c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

// One Philox4x32-10 block.
M355_HD Philox philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 multiply per word pair (v_mad_u64_u32) instead of separate mul_hi / mul_lo
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
    const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    c0 = xor3(hi1, c1, k0);
    c1 = lo1;
    c2 = xor3(hi0, c3, k1);
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox p;
  p.c[0] = c0;
  p.c[1] = c1;
  p.c[2] = c2;
  p.c[3] = c3;
  return p;
}

// Uniform in the open interval (0, 1) from two 32-bit words: the 52 random bits k = (hi >> 12 : lo) become the mantissa
// of a double in [1, 2), d = 1 + k 2^-52, and u = d - (1 - 2^-53) = (k + 1/2) 2^-52.  The difference has 53 significant
// bits, so it is exact, and u lies in [2^-53, 1 - 2^-53]: neither log(u) nor log(1 - u) in the inverse-CDF samplers can
// be infinite.  Two integer operations and one add (the int -> double conversions of both halves cost ten).
M355_HD double u53(unsigned hi, unsigned lo) {
  const unsigned long long bits = ((unsigned long long)(0x3ff00000u | (hi >> 12)) << 32) | (unsigned long long)lo;
  return __builtin_bit_cast(double, bits) - (1.0 - 0x1.0p-53);
}

// Stream of uniform pairs for one element: pair j is Philox block (index, attempt = j, stream).
struct RngState {
  unsigned idx_lo, idx_hi, stream, k0, k1;
  unsigned attempt;
};
M355_HD void next_pair(RngState& s, double& u1, double& u2) {
  const Philox p = philox4x32_10(s.idx_lo, s.idx_hi, s.attempt, s.stream, s.k0, s.k1);
  s.attempt += 1;
  u1 = u53(p.c[0], p.c[1]);
  u2 = u53(p.c[2], p.c[3]);
}

// sqrt(a), a > 0 finite: hardware rsq seed (24 bits) + one coupled Newton step and a residual
// correction (v_sqrt_f64 costs 17 FMAs, this ~8).
M355_DEV double fsqrt(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(a);
  double g = a * y;
  double h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, a);
  return __builtin_fma(d, h, g);
#else
  return ::sqrt(a);
#endif
}

// sin(2 pi u), cos(2 pi u) for u in (0, 1): 4u = n + r exactly (n = rint(4u), |r| <= 1/2), sin / cos of pi r / 2 by two
// degree-7 polynomials in r^2 (tools/gen_sincos_coeffs.py, 1.5e-16), quadrant by swaps and signs.  ~25 operations; the
// device libm's sincospi, with its general argument reduction, was ~60 of the ~95 a normal draw cost.
M355_HD void sincos_2pi(double u, double& s, double& c) {
  const double t = 4.0 * u;
  const double nf = __builtin_rint(t);
  const double r = t - nf;
  const double w = r * r;
  // coefficients in scalar registers (sconst): as literals each one costs a v_mov_b64 per Horner step
  double ps = sconst(kSinP[kSinP_deg]), pc = sconst(kCosP[kCosP_deg]);
#pragma unroll
  for (int k = kSinP_deg - 1; k >= 0; --k) ps = __builtin_fma(ps, w, sconst(kSinP[k]));
#pragma unroll
  for (int k = kCosP_deg - 1; k >= 0; --k) pc = __builtin_fma(pc, w, sconst(kCosP[k]));
  const double sr = r * ps, cr = pc;
  const int q = (int)nf;
  const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

// Two independent standard normals from one uniform pair (Box-Muller).
M355_DEV void normal_pair(double u1, double u2, double& z0, double& z1) {
  const double r = fsqrt(-2.0 * flog_tab(u1));
  double s, c;
  sincos_2pi(u2, s, c);
  z0 = r * c;
  z1 = r * s;
}

constexpr int kMaxAttempts = 256;  // P(no acceptance) < 1e-50 for every sampler below

// cos(2 pi (w + 1/2) 2^-32) for a 32-bit word w: the quadrant is the top two bits, the position in it the other 30 (exact
// integer reduction); cos and sin of the quarter period are ONE polynomial, cos(pi rho / 2) on (0, 1), read forwards or
// backwards (rho -> 1 - rho is the bit complement).  Taylor in rho^2 to (pi/2)^20 / 20! (truncation 1.8e-17 absolute).
M355_HD double cos_2pi_u32(unsigned w) {
  const unsigned q = w >> 30;
  unsigned f = w & 0x3fffffffu;
  if (q & 1u) f = 0x3fffffffu - f;
  const double rho = ((double)f + 0.5) * 0x1.0p-30;
  const double t = rho * rho;
  double p = sconst(3.43773917909860700022e-15);
  p = __builtin_fma(p, t, sconst(-5.29440020073462376193e-13));
  p = __builtin_fma(p, t, sconst(6.56596311497947236221e-11));
  p = __builtin_fma(p, t, sconst(-6.38660308379185224109e-9));
  p = __builtin_fma(p, t, sconst(4.71087477881817150367e-7));
  p = __builtin_fma(p, t, sconst(-0.0000252020423730606054811));
  p = __builtin_fma(p, t, sconst(0.000919260274839426580242));
  p = __builtin_fma(p, t, sconst(-0.0208634807633529608731));
  p = __builtin_fma(p, t, sconst(0.253669507901048013637));
  p = __builtin_fma(p, t, sconst(-1.23370055013616982735));
  p = __builtin_fma(p, t, 1.0);
  return ((q + 1u) & 2u) ? -p : p;
}

// ONE standard normal from three words of a Philox block (single-output Box-Muller): the radius from the 52-bit uniform
// of words 0-1, the angle from the 32 bits of word 2 -- the draw is continuous in the radius, so a 2^-32 grid of angles
// costs no resolution -- which leaves word 3 to the caller (the acceptance uniform of a rejection sampler).
M355_DEV double normal_single(const Philox& p) {
  return fsqrt(-2.0 * flog_tab(u53(p.c[0], p.c[1]))) * cos_2pi_u32(p.c[2]);
}

// Standard normal by the ziggurat (Marsaglia & Tsang 2000; 1024 layers, tables of m355_zigtab.h) -- numpy's Generator, which
// the reference reaches through pt.random.normal (normal.py:66-67), samples the same way.  64 random bits per candidate:
// bit 31 of `hi` is the sign, the ten bits below it the layer, and the 52 bits (hi & 0xfffff : lo) are the mantissa of d = 1 + u in
// [1, 2); the candidate u x_i is accepted at once if d is below the layer's threshold 1 + x_(i+1) / x_i (99.6 % of the
// candidates: one table read, one compare, one fma).  Otherwise zig_slow decides: the wedge of the other layers by
// log(y) < -x^2 / 2 with y uniform between f(x_i) and f(x_(i+1)), the tail of layer 0 by Marsaglia's exponential
// rejection.  A rejected candidate means a whole new attempt (new layer).
// Counters: attempt 0 of elements 2k and 2k+1 is the two halves of block (k, 0); attempt a >= 1 of element g is words
// 0-1 of block (g, a << 8); the extra uniforms of attempt a of element g come from blocks (g, (a << 8) + 1 + t), t the
// iteration of the tail loop -- blocks with a non-zero third word belong to one element only.
constexpr int kZigBits = kZigLayers == 256 ? 8 : kZigLayers == 512 ? 9 : kZigLayers == 1024 ? 10 : 11;
static_assert(kZigLayers == (1 << kZigBits), "the layer is a bit field of the candidate's high word");
M355_HD int zig_layer(unsigned hi) { return (int)((hi >> (31 - kZigBits)) & (unsigned)(kZigLayers - 1)); }
M355_HD bool zig_fast(unsigned hi, unsigned lo, double thr, double xi, double& x) {
  const double d = __builtin_bit_cast(double, ((unsigned long long)((hi & 0x000fffffu) | 0x3ff00000u) << 32) | (unsigned long long)lo);
  const unsigned long long mag = __builtin_bit_cast(unsigned long long, __builtin_fma(d, xi, -xi));
  x = __builtin_bit_cast(double, ((unsigned long long)((hi & 0x80000000u) | (unsigned)(mag >> 32)) << 32) | (mag & 0xffffffffull));
  return d < thr;
}
struct ZigCtr {
  unsigned long long g;  // global element index
  unsigned stream, k0, k1;
};
M355_HD void zig_bits(const ZigCtr& z, unsigned a, unsigned& hi, unsigned& lo) {
  if (a == 0u) {
    const unsigned long long c = z.g >> 1;
    const Philox p = philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, z.stream, z.k0, z.k1);
    hi = (z.g & 1ull) ? p.c[2] : p.c[0];
    lo = (z.g & 1ull) ? p.c[3] : p.c[1];
  } else {
    const Philox p = philox4x32_10((unsigned)z.g, (unsigned)(z.g >> 32), a << 8, z.stream, z.k0, z.k1);
    hi = p.c[0];
    lo = p.c[1];
  }
}
// the candidate x of layer `layer` failed the threshold test: accept (true; the tail replaces x) or reject
M355_DEV bool zig_slow(const ZigCtr& z, unsigned a, int layer, double& x) {
  if (layer == 0) {
    for (unsigned t = 0;; ++t) {
      const Philox e = philox4x32_10((unsigned)z.g, (unsigned)(z.g >> 32), (a << 8) + 1u + t, z.stream, z.k0, z.k1);
      const double xx = -flog(u53(e.c[0], e.c[1])) * kZigRInv;
      const double yy = -flog(u53(e.c[2], e.c[3]));
      if (yy + yy > xx * xx || t >= 253u) {
        x = __builtin_signbit(x) ? -(kZigR + xx) : kZigR + xx;
        return true;
      }
    }
  }
  const Philox e = philox4x32_10((unsigned)z.g, (unsigned)(z.g >> 32), (a << 8) + 1u, z.stream, z.k0, z.k1);
  const double f0 = kZigF[layer], f1 = kZigF[layer + 1];
  const double y = __builtin_fma(u53(e.c[0], e.c[1]), f1 - f0, f0);
  return flog(y) < -0.5 * x * x;
}
// per-lane loop (k_rvs and the host build); k_rvs_normal_zig runs the same attempts from a queue
M355_DEV double sample_std_normal(const ZigCtr& z) {
  double x = 0.0;
  for (unsigned a = 0; a < (unsigned)kMaxAttempts; ++a) {
    unsigned hi, lo;
    zig_bits(z, a, hi, lo);
    const int layer = zig_layer(hi);
    if (zig_fast(hi, lo, kZigTab[2 * layer], kZigTab[2 * layer + 1], x)) break;
    if (zig_slow(z, a, layer, x)) break;
  }
  return x;
}

// Uniform in (0, 1) on a 2^-44 grid from word 3 and the twelve low bits of word 0 that u53 drops (mantissa trick as in
// u53; the forced low bit keeps it off 0: log(u) is finite).
M355_HD double u44(unsigned w3, unsigned w0) {
  const unsigned long long bits = ((unsigned long long)(0x3ff00000u | (w3 >> 12)) << 32) |
                                  (unsigned long long)((w3 << 20) | ((w0 & 0xfffu) << 8) | 0x80u);
  return __builtin_bit_cast(double, bits) - 1.0;
}

// Gamma(shape a >= 1, scale 1), Marsaglia & Tsang (2000): the shape-only constants and ONE attempt from ONE Philox block
// (normal from words 0-2, acceptance uniform from word 3).  Shared by the per-lane loop (sample_std_gamma) and the
// compacting kernel (k_rvs_mt), so both produce the same draw for the same (element, attempt) counters.
struct MtK {
  double d, c;
};
M355_DEV MtK mt_setup(double a) {
  MtK k;
  k.d = a - 1.0 / 3.0;
  k.c = 1.0 / (3.0 * fsqrt(k.d));
  return k;
}
// WANT_Z: also the second Box-Muller output of the same block, z = r sin(angle) -- a standard normal independent of x and
// of the acceptance uniform, hence of the accepted gamma value: Student t takes its numerator from it (one block per
// attempt for the whole draw, no separate normal).  sin(2 pi w) is the cosine a quarter period earlier: w - 2^30.
template <bool WANT_Z = false> M355_DEV bool mt_attempt(const MtK& k, const Philox& p, double& g, double& x, double* z = nullptr) {
  if constexpr (WANT_Z) {
    const double r = fsqrt(-2.0 * flog_tab(u53(p.c[0], p.c[1])));
    x = r * cos_2pi_u32(p.c[2]);
    *z = r * cos_2pi_u32(p.c[2] - 0x40000000u);
  } else {
    x = normal_single(p);
  }
  const double u = u44(p.c[3], p.c[0]);
  const double t = 1.0 + k.c * x;
  const double vv = t * t * t;
  const double x2 = x * x;
  g = k.d * vv;
  if (!(t > 0.0)) return false;
  if (u < 1.0 - 0.0331 * x2 * x2) return true;
  // second squeezes with the shape in them: the exact test is log u < d S(s), s = c x, S(s) = sum_{k >= 4} 3 (-1)^(k+1) s^k / k
  // (the terms below s^4 of x^2 / 2 + d (1 - v + log v) cancel because d c^2 = 1/9).  On |s| <= 3/4:
  //   S(s) >= -3/4 s^4 + 3/5 s^5 - 1/2 s^6 - 12/7 |s|^7  and  log u <= u - 1      -> accept if u - 1 < d * lower;
  //   S(s) <= -3/4 s^4 (+ 3/5 s^5 for s > 0)             and  log u >= (u - 1) / u -> reject if u - 1 >= u d * upper.
  // Undecided after these: 0.9 % of the attempts at shape 2.5, 0.05 % at 10 (Marsaglia & Tsang's shape-free
  // 0.0331 x^4 alone leaves 8 %, i.e. some lane of a wavefront in every round); the outcome of an attempt is unchanged.
  const double sx = k.c * x, s2 = sx * sx, s4 = s2 * s2;
  if (k.d >= 1.5 && __builtin_fabs(sx) <= 0.75) {  // (below shape ~1.8 they decide too little to pay for themselves)
    const double lower = s4 * (-0.75 + sx * (0.6 - 0.5 * sx)) - (12.0 / 7.0) * (s4 * s2) * __builtin_fabs(sx);
    const double upper = s4 * (-0.75 + (sx > 0.0 ? 0.6 * sx : 0.0));
    if (u - 1.0 < k.d * lower) return true;
    if (u - 1.0 >= u * k.d * upper) return false;
  }
  return flog_tab(u) < 0.5 * x2 + k.d * (1.0 - vv + flog_tab(vv));
}

// Gamma(shape a, scale 1).  a == 0 -> 0; a < 0 or NaN -> NaN.  a < 1: Gamma(a + 1) U^(1/a), one more block.
// zout: receives the accepted attempt's second normal (mt_attempt<true>); left alone for a <= 0, NaN or infinite a.
template <bool WANT_Z = false> M355_DEV double sample_std_gamma(RngState& st, double a, double* zout = nullptr) {
  if (!(a >= 0.0)) return kNaN;
  if (a == 0.0) return 0.0;
  if (a == kInf) return kInf;
  const bool boost = a < 1.0;
  const MtK k = mt_setup(boost ? a + 1.0 : a);
  double g = k.d;
  for (int it = 0; it < kMaxAttempts; ++it) {
    const Philox p = philox4x32_10(st.idx_lo, st.idx_hi, st.attempt, st.stream, st.k0, st.k1);
    st.attempt += 1;
    double x;
    if (mt_attempt<WANT_Z>(k, p, g, x, zout)) break;
  }
  if (boost) {
    double u1, u2;
    next_pair(st, u1, u2);
    g *= exp_(flog_tab(u1) / a);
  }
  return g;
}

// Beta(a0, b0) with both shapes above 1: Cheng's algorithm BB (1978; R's rbeta uses it) -- ONE attempt from one Philox
// block (two uniforms), a logistic proposal for log(X / (1 - X)) with two squeeze steps in front of the exact test;
// expected attempts <= 4/e = 1.47 at a, b -> 1 and ~1.1 for larger shapes, where the gamma ratio of numpy's Generator
// (the reference's sampler, beta.py:86-87) costs two Marsaglia-Tsang draws.  Shared by the per-lane loop (sample_beta_bb)
// and the compacting kernel (k_rvs_bb).
struct BbK {
  double a, b, alpha, beta, gamma;  // a = min(a0, b0), b = max(a0, b0)
  bool swapped;                     // a0 > b0: the draw is b / (b + w) instead of w / (b + w)
};
M355_HD bool bb_applies(double a0, double b0) { return a0 > 1.0 && b0 > 1.0 && a0 < kInf && b0 < kInf; }
M355_DEV BbK bb_setup(double a0, double b0) {
  BbK k;
  k.swapped = a0 > b0;
  k.a = k.swapped ? b0 : a0;
  k.b = k.swapped ? a0 : b0;
  k.alpha = k.a + k.b;
  k.beta = fsqrt((k.alpha - 2.0) / (2.0 * k.a * k.b - k.alpha));
  k.gamma = k.a + 1.0 / k.beta;
  return k;
}
M355_DEV bool bb_attempt(const BbK& k, const Philox& p, double& x) {
  const double u1 = u53(p.c[0], p.c[1]), u2 = u53(p.c[2], p.c[3]);
  const double v = k.beta * flog_tab(u1 / (1.0 - u1));
  double w = k.a * exp_(v);
  w = w < 1.7976931348623157e308 ? w : 1.7976931348623157e308;
  const double z = u1 * u1 * u2;
  const double r = k.gamma * v - 1.3862943611198906;  // log 4
  const double s = k.a + r - w;
  x = k.swapped ? k.b / (k.b + w) : w / (k.b + w);
  if (s + 2.6094379124341005 >= 5.0 * z) return true;  // 1 + log 5
  const double t = flog_tab(z);
  if (s > t) return true;
  return r + k.alpha * flog_tab(k.alpha / (k.b + w)) >= t;
}
M355_DEV double sample_beta_bb(RngState& st, double a0, double b0) {
  const BbK k = bb_setup(a0, b0);
  double x = 0.5;
  for (int it = 0; it < kMaxAttempts; ++it) {
    const Philox p = philox4x32_10(st.idx_lo, st.idx_hi, st.attempt, st.stream, st.k0, st.k1);
    st.attempt += 1;
    if (bb_attempt(k, p, x)) break;
  }
  return x;
}

// log(k!) for integer-valued k >= 0: table below 16, Stirling's series above.
M355_DEV double lfactorial(double k) {
  if (k < 16.0) {
    const double t[16] = {0.0,
                          0.0,
                          0.69314718055994530942,
                          1.79175946922805500081,
                          3.17805383034794561965,
                          4.78749174278204599425,
                          6.57925121201010099506,
                          8.52516136106541430017,
                          10.60460290274525022842,
                          12.80182748008146961121,
                          15.10441257307551529523,
                          17.50230784587388583929,
                          19.98721449566188614952,
                          22.55216385312342288557,
                          25.19122118273868150009,
                          27.89927138384089156609};
    return t[(int)k];
  }
  const double z = k + 1.0;
  return (z - 0.5) * flog_tab(z) - z + kLogSqrt2Pi + stirling_corr(z);
}

// PTRS (Hoermann 1993, "The transformed rejection method for generating Poisson random
// variables"), mu >= 10: the mu-only constants and one attempt.  Shared by the per-lane loop
// (sample_poisson) and the compacting kernel (k_rvs_compact), so both produce the same draw for the
// same (element, attempt) counters.
struct PtrsK {
  double mu, loglam, b, a, vr, log_invalpha;
};
M355_DEV PtrsK ptrs_setup(double mu) {
  PtrsK c;
  const double slam = fsqrt(mu);
  c.mu = mu;
  c.loglam = flog_tab(mu);
  c.b = 0.931 + 2.53 * slam;
  c.a = -0.059 + 0.02483 * c.b;
  c.log_invalpha = flog_tab(1.1239 + 1.1328 / (c.b - 3.4));
  c.vr = 0.9277 - 3.6224 / (c.b - 2.0);
  return c;
}
// One PTRS attempt from the uniform pair (u1, v), in two parts: the squeeze tests (1 = accepted, 0 = rejected, 2 = the
// logarithmic test has to decide; k holds the candidate, us the distance of u from the ends) and that test.  The
// compacting kernel queues the undecided attempts of a wavefront and runs the logarithms on 64 of them at a time.
M355_DEV int ptrs_fast(const PtrsK& c, double u1, double v, double& k, double& us) {
  const double u = u1 - 0.5;
  us = 0.5 - __builtin_fabs(u);
  k = floor_((2.0 * c.a / us + c.b) * u + c.mu + 0.43);
  if (us >= 0.07 && v <= c.vr) return 1;
  if (k < 0.0 || (us < 0.013 && v > us)) return 0;
  return 2;
}
M355_DEV bool ptrs_slow(const PtrsK& c, double k, double us, double v) {
  return flog_tab(v) + c.log_invalpha - flog_tab(c.a / (us * us) + c.b) <= -c.mu + k * c.loglam - lfactorial(k);
}
// true = accepted, k holds the draw.
M355_DEV bool ptrs_attempt(const PtrsK& c, double u1, double v, double& k) {
  double us;
  const int f = ptrs_fast(c, u1, v, k, us);
  return f == 2 ? ptrs_slow(c, k, us, v) : f == 1;
}

// Poisson(mu) as a double holding an integer.  mu == 0 -> 0; mu < 0 or NaN -> NaN.
M355_DEV double sample_poisson(RngState& st, double mu) {
  if (!(mu >= 0.0)) return kNaN;
  if (mu == 0.0) return 0.0;
  if (mu < 10.0) {
    // sequential inversion from one uniform
    double u1, u2;
    next_pair(st, u1, u2);
    double p = exp_(-mu), s = p, k = 0.0;
    while (u1 > s && k < 200.0) {
      k += 1.0;
      p *= mu / k;
      s += p;
    }
    return k;
  }
  const PtrsK c = ptrs_setup(mu);
  double k = floor_(mu);
  for (int it = 0; it < kMaxAttempts; ++it) {
    double u1, v;
    next_pair(st, u1, v);
    if (ptrs_attempt(c, u1, v, k)) break;
  }
  return k;
}

// Stirling tail log(k!) - [ (k+1/2) log(k+1) - (k+1) + log(2 pi)/2 ] used by BTRS.
M355_DEV double stirling_tail(double k) {
  if (k < 10.0) {
    const double t[10] = {0.0810614667953272,  0.0413406959554092,  0.0276779256849983,  0.02079067210376509,
                          0.0166446911898211,  0.0138761288230707,  0.0118967099458917,  0.0104112652619720,
                          0.00925546218271273, 0.00833056343336287};
    return t[(int)k];
  }
  const double kp1 = k + 1.0, kp1sq = kp1 * kp1;
  return (1.0 / 12.0 - (1.0 / 360.0 - 1.0 / 1260.0 / kp1sq) / kp1sq) / kp1;
}

// BTRS (Hoermann 1993, "The generation of binomial random variates"), n*min(p,1-p) >= 10: the
// parameter-only constants and one attempt (shared like PTRS above).
struct BtrsK {
  double n, r, b, a, c, vr, alpha, m, hm;
  bool flip;
};
M355_DEV BtrsK btrs_setup(double n, double p) {
  BtrsK k;
  k.flip = p > 0.5;
  const double pp = k.flip ? 1.0 - p : p;
  const double q = 1.0 - pp;
  const double spq = fsqrt(n * pp * q);
  k.n = n;
  k.b = 1.15 + 2.53 * spq;
  k.a = -0.0873 + 0.0248 * k.b + 0.01 * pp;
  k.c = n * pp + 0.5;
  k.vr = 0.92 - 4.2 / k.b;
  k.r = pp / q;
  k.alpha = (2.83 + 5.1 / k.b) * spq;
  k.m = floor_((n + 1.0) * pp);
  // the k-independent part of the acceptance bound
  k.hm = (k.m + 0.5) * flog_tab((k.m + 1.0) / (k.r * (n - k.m + 1.0))) + stirling_tail(k.m) + stirling_tail(n - k.m);
  return k;
}
M355_DEV int btrs_fast(const BtrsK& c, double u1, double v, double& k, double& us) {  // (as ptrs_fast)
  const double u = u1 - 0.5;
  us = 0.5 - __builtin_fabs(u);
  k = floor_((2.0 * c.a / us + c.b) * u + c.c);
  if (us >= 0.07 && v <= c.vr) return 1;
  if (k < 0.0 || k > c.n) return 0;
  return 2;
}
M355_DEV bool btrs_slow(const BtrsK& c, double k, double us, double v) {
  const double lv = flog_tab(v * c.alpha / (c.a / (us * us) + c.b));
  const double upper = c.hm + (c.n + 1.0) * flog_tab((c.n - c.m + 1.0) / (c.n - k + 1.0)) +
                       (k + 0.5) * flog_tab(c.r * (c.n - k + 1.0) / (k + 1.0)) - stirling_tail(k) - stirling_tail(c.n - k);
  return lv <= upper;
}
M355_DEV bool btrs_attempt(const BtrsK& c, double u1, double v, double& k) {
  double us;
  const int f = btrs_fast(c, u1, v, k, us);
  return f == 2 ? btrs_slow(c, k, us, v) : f == 1;
}

// Binomial(n, p) as a double holding an integer.  n must be a non-negative integer value.
M355_DEV double sample_binomial(RngState& st, double n, double p) {
  if (!(n >= 0.0) || !(p >= 0.0) || !(p <= 1.0)) return kNaN;
  if (n == 0.0 || p == 0.0) return 0.0;
  if (p == 1.0) return n;
  const bool flip = p > 0.5;
  const double pp = flip ? 1.0 - p : p;
  double k = 0.0;
  if (n * pp < 10.0) {
    // sequential inversion from one uniform (restarts in the ~1e-16 tail the sum cannot reach)
    const double q = 1.0 - pp;
    const double r = pp / q;
    for (int it = 0; it < kMaxAttempts; ++it) {
      double u1, u2;
      next_pair(st, u1, u2);
      double px = exp_(n * flog1p(-pp)), s = px;
      k = 0.0;
      while (u1 > s && k < n) {
        k += 1.0;
        px *= r * (n - k + 1.0) / k;
        s += px;
      }
      if (u1 <= s || k >= n) break;
    }
  } else {
    const BtrsK c = btrs_setup(n, p);
    k = c.m;
    for (int it = 0; it < kMaxAttempts; ++it) {
      double u1, v;
      next_pair(st, u1, v);
      if (btrs_attempt(c, u1, v, k)) break;
    }
  }
  return flip ? n - k : k;
}

// ---- Discrete families with scalar parameters: inversion on the tabulated cdf -------------------------------------------
// With one parameter set per call the cdf is a function of an integer in the window that Dist<>::dtab_range proposes (the
// same window k_dtab uses): c[i] = cdf(lo + i), evaluated once per workgroup by the exact body, and a draw is the smallest
// i with c[i] >= u -- the family's own ppf at a uniform u (discreteweibull.py:78-79 samples that way in the reference;
// numpy's samplers for the others are rejection methods with the same law).  A guide table makes the search O(1): g[j] is
// the answer for u = j / 4096, and the twelve bits of u that select j are the top bits of the Philox word u is made of, so
// the search runs between two neighbouring guide entries.  64 random bits per draw: elements 2k and 2k + 1 take the two
// halves of block (k, 0), like the normal sampler.  Resolution: u lies on a 2^-53 grid in (0, 1), so atoms beyond
// sf < 1.1e-16 are never drawn (PTRS / BTRS stop at the same scale: their acceptance tests use the same uniforms).
constexpr int kInvGuideBits = 12;
constexpr int kInvGuide = 1 << kInvGuideBits;
M355_HD int inv_guide(const double* c, int len, int j) {  // smallest i with c[i] >= j / kInvGuide (len - 1 if there is none)
  const double t = (double)j * (1.0 / (double)kInvGuide);
  int lo = 0, hi = len - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (c[mid] >= t) hi = mid;
    else lo = mid + 1;
  }
  while (lo > 0 && c[lo - 1] >= t) --lo;  // (the exact body may wobble in the last bit where the cdf is flat)
  return lo;
}
// g has kInvGuide + 1 entries (the last one len - 1): the answer for u in cell j lies in [g[j], g[j + 1]] -- the same index
// in most cells, one apart where an atom's boundary crosses the cell (one comparison decides), further apart only where
// atoms are lighter than a cell, i.e. in the tails (the cell u < 2^-12 covers the whole lower tail): bisection there
M355_HD int inv_index(const double* c, const unsigned short* g, unsigned hi, double u) {
  const unsigned j = hi >> (32 - kInvGuideBits);
  int lo = (int)g[j], up = (int)g[j + 1u];
  if (up - lo > 1) {
    while (lo + 1 < up) {  // invariant: the answer is in [lo, up]
      const int mid = (lo + up) >> 1;
      if (c[mid] >= u) up = mid;
      else lo = mid + 1;
    }
  }
  return c[lo] < u ? up : lo;  // (up == lo: c[lo] >= u, or the window ends below u -- inv_draw's exact continuation)
}
// u above the last tabulated value (the window ends where sf ~ 1e-40: only rounding of c[len - 1] leads here): go on
// with the exact body
template <int D> M355_DEV_HEAVY double inv_beyond(double v, double u, const typename Dist<D>::K& k) {
  for (int step = 0; step < 4096 && Dist<D>::dtab_cdf(v, k) < u; ++step) v += 1.0;
  return v;
}
// the families sampled this way, and the parameter sets they take (anything else: the per-lane samplers of sample_one)
template <int D> struct InvFamily { static constexpr bool value = false; };
#define M355_INV_FAMILY(ID, COND)                                  \
  template <> struct InvFamily<ID> {                               \
    static constexpr bool value = true;                            \
    static M355_HD bool ok(const double* p) { return COND; }       \
  };
M355_HD bool inv_count(double n) { return n >= 1.0 && n < kInf && n == __builtin_floor(n); }
M355_HD bool inv_prob(double p) { return p > 0.0 && p < 1.0; }
M355_INV_FAMILY(M355_POISSON, p[0] > 0.0 && p[0] < kInf)
M355_INV_FAMILY(M355_BINOMIAL, inv_count(p[0]) && inv_prob(p[1]))
M355_INV_FAMILY(M355_NEGATIVEBINOMIAL, p[0] > 0.0 && p[0] < kInf && inv_prob(p[1]))
M355_INV_FAMILY(M355_BETABINOMIAL, inv_count(p[0]) && p[1] > 0.0 && p[1] < kInf && p[2] > 0.0 && p[2] < kInf)
M355_INV_FAMILY(M355_HYPERGEOMETRIC, inv_count(p[0]) && inv_count(p[1]) && inv_count(p[2]) && p[1] <= p[0] && p[2] <= p[0])
M355_INV_FAMILY(M355_ZI_POISSON, p[0] > 0.0 && p[0] <= 1.0 && p[1] > 0.0 && p[1] < kInf)
M355_INV_FAMILY(M355_ZI_BINOMIAL, p[0] > 0.0 && p[0] <= 1.0 && inv_count(p[1]) && inv_prob(p[2]))
M355_INV_FAMILY(M355_ZI_NEGATIVEBINOMIAL, p[0] > 0.0 && p[0] <= 1.0 && p[1] > 0.0 && p[1] < kInf && inv_prob(p[2]))
#undef M355_INV_FAMILY
#define M355_INV_LIST(X)                                                                                        \
  X(M355_POISSON) X(M355_BINOMIAL) X(M355_NEGATIVEBINOMIAL) X(M355_BETABINOMIAL) X(M355_HYPERGEOMETRIC) \
  X(M355_ZI_POISSON) X(M355_ZI_BINOMIAL) X(M355_ZI_NEGATIVEBINOMIAL)

template <int D>
M355_DEV double inv_draw(const double* c, const unsigned short* g, int len, double lo, unsigned hi, unsigned lw,
                         const typename Dist<D>::K& k) {
  const double u = u53(hi, lw);
  const int i = inv_index(c, g, hi, u);
  const double v = lo + (double)i;
  if (i == len - 1 && c[i] < u) return inv_beyond<D>(v, u, k);
  return v;
}

// Families whose draw is a function of ONE uniform (inverse-cdf samplers): elements 2k and 2k + 1 take the two halves of
// block (k, 0), so a thread of k_rvs_pairs turns one Philox block into two draws and one 16-byte store.
template <int D> constexpr bool one_uniform() {
  return D == M355_EXPONENTIAL || D == M355_CAUCHY || D == M355_HALFCAUCHY || D == M355_LAPLACE || D == M355_LOGISTIC ||
         D == M355_GUMBEL || D == M355_WEIBULL || D == M355_PARETO || D == M355_UNIFORM || D == M355_TRIANGULAR ||
         D == M355_KUMARASWAMY || D == M355_ASYMMETRICLAPLACE || D == M355_LOGLOGISTIC || D == M355_BERNOULLI ||
         D == M355_GEOMETRIC || D == M355_DISCRETEUNIFORM || D == M355_DISCRETEWEIBULL || D == M355_TRUNCATEDNORMAL ||
         D == M355_SKEW_STUDENTT;
}
template <int D> M355_DEV double from_uniform(double u1, const double* p) {
  if constexpr (D == M355_BERNOULLI) {  // bernoulli.py:41-42: binomial(1, p)
    return (u1 < p[0]) ? 1.0 : 0.0;
  }
  if constexpr (D == M355_GEOMETRIC) {  // geometric.py:63-64: inversion, support {1, 2, ...}
    const double g = __builtin_ceil(flog_tab(u1) / flog1p(-p[0]));
    return g < 1.0 ? 1.0 : g;
  }
  if constexpr (D == M355_DISCRETEUNIFORM) {  // discreteuniform.py:64-65: integers(lower, upper + 1)
    const double n = p[1] - p[0] + 1.0;
    const double j = floor_(u1 * n);
    return p[0] + (j < n ? j : n - 1.0);
  }
  if constexpr (D == M355_DISCRETEWEIBULL) {  // discreteweibull.py:78-79: ppf(uniform)
    const double v = __builtin_ceil(pow_(flog(1.0 - u1) / flog(p[0]), 1.0 / p[1]) - 1.0);
    return v < 0.0 ? 0.0 : v;
  }
  if constexpr (D == M355_TRUNCATEDNORMAL || D == M355_SKEW_STUDENTT) {
    // truncatednormal.py:273-275, skew_studentt.py:164-166: ppf(uniform(0, 1))
    return Dist<D>::template eval<M355_PPF>(u1, Dist<D>::derive(p));
  }
  if constexpr (D == M355_EXPONENTIAL) return -flog_tab(u1) / p[0];  // exponential.py:62-63
  if constexpr (D == M355_CAUCHY || D == M355_HALFCAUCHY) {  // cauchy.py:62-63, halfcauchy.py:69-71 (scale is p[0] there)
    // tan(pi (u - 1/2)) and tan(pi u / 2) as sin / cos of sincos_2pi: u - 1/2 is exact, the reduced argument is small where
    // the tangent is large (the cosine is then r * polynomial: full relative accuracy in the tails) -- a third of the cost of
    // the device libm's tan with its general argument reduction
    double sn, cs;
    sincos_2pi(D == M355_CAUCHY ? 0.5 * (u1 - 0.5) : 0.25 * u1, sn, cs);
    const double t = sn / cs;
    return D == M355_CAUCHY ? p[0] + p[1] * t : p[0] * t;
  }
  if constexpr (D == M355_LAPLACE)                                                      // laplace.py:68-69
    return (u1 < 0.5) ? p[0] + p[1] * flog_tab(2.0 * u1) : p[0] - p[1] * flog_tab(2.0 * (1.0 - u1));
  if constexpr (D == M355_LOGISTIC) return p[0] + p[1] * flog_tab(u1 / (1.0 - u1));  // logistic.py:58-59
  if constexpr (D == M355_GUMBEL) return p[0] - p[1] * flog_tab(-flog_tab(u1));  // gumbel.py:61-62
  if constexpr (D == M355_WEIBULL) return p[1] * exp_(flog_tab(-flog_tab(u1)) / p[0]);  // weibull.py:64-65
  if constexpr (D == M355_PARETO) return p[1] * exp_(-flog_tab(u1) / p[0]);                    // pareto.py:63-65
  if constexpr (D == M355_UNIFORM) return p[0] + u1 * (p[1] - p[0]);                     // uniform.py:66-67
  if constexpr (D == M355_TRIANGULAR) {                                                 // triangular.py:85-87 (own ppf)
    const double w = p[2] - p[0];
    return (u1 < (p[1] - p[0]) / w) ? p[0] + sqrt_(w * (p[1] - p[0]) * u1) : p[2] - sqrt_(w * (p[2] - p[1]) * (1.0 - u1));
  }
  if constexpr (D == M355_KUMARASWAMY)                                                  // kumaraswamy.py:69-71 (own ppf)
    return pow_(1.0 - pow_(1.0 - u1, 1.0 / p[1]), 1.0 / p[0]);
  if constexpr (D == M355_ASYMMETRICLAPLACE) {                                          // asymmetriclaplace.py:91-96
    const double r = -p[2] + u1 * (1.0 / p[2] + p[2]);  // uniform(-kappa, 1/kappa)
    const double sgn = (r > 0.0) ? 1.0 : ((r < 0.0) ? -1.0 : 0.0);
    const double ks = pow_(p[2], sgn);
    return p[0] - (1.0 / (1.0 / p[1] * sgn * ks)) * flog(1.0 - r * sgn * ks);
  }
  if constexpr (D == M355_LOGLOGISTIC) return p[0] * exp_(flog_tab(u1 / (1.0 - u1)) / p[1]);  // loglogistic.py:107-109
  return kNaN;
}

// ... and the families that are a function of ONE standard normal: elements 2k and 2k + 1 take the cosine and the sine
// branch of the Box-Muller pair of block (k, 0).
template <int D> constexpr bool one_normal() {
  return D == M355_LOGNORMAL || D == M355_HALFNORMAL || D == M355_MOYAL || D == M355_LOGITNORMAL;
}
template <int D> M355_DEV double from_normal(double z, const double* p) {
  if constexpr (D == M355_LOGNORMAL) return exp_(p[0] + p[1] * z);       // lognormal.py:68-69
  if constexpr (D == M355_HALFNORMAL) return __builtin_fabs(p[0] * z);   // halfnormal.py:62-63
  if constexpr (D == M355_MOYAL) return p[0] - p[1] * flog_tab(z * z);  // moyal.py:66-68: ppf(U) = mu - sigma log Z^2
  if constexpr (D == M355_LOGITNORMAL) return sigmoid_(p[0] + p[1] * z);  // logitnormal.py:191-192
  return kNaN;
}

// von Mises by Best & Fisher's wrapped-Cauchy rejection (numpy's random_vonmises), kappa >= 1e-8: the kappa-only constant,
// one attempt from a uniform pair, and the last step (sign from another uniform, shift by mu, wrap into [-pi, pi]).  Shared
// by the per-lane loop (sample_one) and the compacting kernel (k_rvs_vm).
M355_DEV double vm_setup(double kappa) {
  if (kappa < 1e-5) return 1.0 / kappa + kappa;
  const double r = 1.0 + fsqrt(1.0 + 4.0 * kappa * kappa);
  const double rho = (r - fsqrt(2.0 * r)) / (2.0 * kappa);
  return (1.0 + rho * rho) / (2.0 * rho);
}
M355_DEV bool vm_attempt(double s, double kappa, double u1, double u2, double& w) {
  double sn, z;
  sincos_2pi(0.5 * u1, sn, z);  // z = cos(pi u1) (the device libm's cos, with its general reduction, cost 2.5 times that)
  w = (1.0 + s * z) / (s + z);
  const double y = kappa * (s - w);
  return (y * (2.0 - y) - u2 >= 0.0) || (flog_tab(y / u2) + 1.0 - y >= 0.0);
}
M355_DEV double vm_finish(double w, double u1, double mu) {
  double res = ::acos(w);
  if (u1 < 0.5) res = -res;
  res += mu;
  const bool neg = res < 0.0;
  double m = __builtin_fabs(res) + kPi;
  // fmod(m, 2 pi): below 4 pi it is m or m - 2 pi, and that subtraction is exact -- the same value without fmod's loop
  if (m >= 4.0 * kPi) m = ::fmod(m, 2.0 * kPi);
  else if (m >= 2.0 * kPi) m -= 2.0 * kPi;
  m -= kPi;
  return neg ? -m : m;
}

// One draw of distribution D for global element index `idx` (parameters already fetched).
template <int D> M355_DEV double sample_one(unsigned long long idx, unsigned long long seed, unsigned stream,
                                           const double* p) {
  RngState st;
  st.idx_lo = (unsigned)idx;
  st.idx_hi = (unsigned)(idx >> 32);
  st.stream = stream;
  st.k0 = (unsigned)seed;
  st.k1 = (unsigned)(seed >> 32);
  st.attempt = 0;
  if constexpr (one_uniform<D>()) {
    const unsigned long long c = idx >> 1;
    const Philox b = philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, stream, st.k0, st.k1);
    return from_uniform<D>((idx & 1ull) ? u53(b.c[2], b.c[3]) : u53(b.c[0], b.c[1]), p);
  } else if constexpr (one_normal<D>()) {
    const unsigned long long c = idx >> 1;
    const Philox b = philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, stream, st.k0, st.k1);
    double z0, z1;
    normal_pair(u53(b.c[0], b.c[1]), u53(b.c[2], b.c[3]), z0, z1);
    return from_normal<D>((idx & 1ull) ? z1 : z0, p);
  } else if constexpr (D == M355_NORMAL) {  // normal.py:66-67: mu + sigma * z
    const ZigCtr z{idx, stream, st.k0, st.k1};
    return __builtin_fma(p[1], sample_std_normal(z), p[0]);
  } else if constexpr (D == M355_GAMMA) {  // gamma.py:61-62: shape alpha, scale 1/beta
    return sample_std_gamma(st, p[0]) / p[1];
  } else if constexpr (D == M355_BETA) {  // beta.py:86-87
    if (bb_applies(p[0], p[1])) return sample_beta_bb(st, p[0], p[1]);
    const double ga = sample_std_gamma(st, p[0]);
    const double gb = sample_std_gamma(st, p[1]);
    return ga / (ga + gb);
  } else if constexpr (D == M355_STUDENTT) {  // studentt.py:96-97: mu + sigma * z / sqrt(chi2_nu / nu)
    // the numerator is the second Box-Muller output of the accepted gamma attempt's block (mt_attempt<true>); where the
    // gamma leg makes no attempt (nu = 0, infinite or NaN) a block of its own
    const double a = 0.5 * p[0];
    double z0 = 0.0;
    if (!(a > 0.0) || a == kInf) z0 = normal_single(philox4x32_10(st.idx_lo, st.idx_hi, 0x7fffffffu, st.stream, st.k0, st.k1));
    const double g = sample_std_gamma<true>(st, a, &z0);  // chi2_nu / 2
    return p[1] + p[2] * z0 * fsqrt(a / g);
  } else if constexpr (D == M355_POISSON) {  // poisson.py:66-67
    return sample_poisson(st, p[0]);
  } else if constexpr (D == M355_BINOMIAL) {  // binomial.py:62-63
    return sample_binomial(st, p[0], p[1]);
  } else if constexpr (D == M355_NEGATIVEBINOMIAL) {  // negativebinomial.py:66-67: Poisson(Gamma(n, (1-p)/p))
    const double lam = sample_std_gamma(st, p[0]) * ((1.0 - p[1]) / p[1]);
    return sample_poisson(st, lam);
  } else if constexpr (D == M355_BETABINOMIAL) {  // betabinomial.py:111-112: binomial(n, Beta(alpha, beta))
    const double ga = sample_std_gamma(st, p[1]);
    const double gb = sample_std_gamma(st, p[2]);
    return sample_binomial(st, p[0], ga / (ga + gb));
  } else if constexpr (D == M355_HYPERGEOMETRIC) {  // hypergeometric.py:115-116: n draws without replacement
    double good = p[1], total = p[0], x = 0.0;
    const int draws = (int)p[2];
    for (int i = 0; i < draws; i += 2) {
      double u1, u2;
      next_pair(st, u1, u2);
      if (u1 * total < good) { x += 1.0; good -= 1.0; }
      total -= 1.0;
      if (i + 1 < draws) {
        if (u2 * total < good) { x += 1.0; good -= 1.0; }
        total -= 1.0;
      }
    }
    return x;
  } else if constexpr (D == M355_ZI_POISSON || D == M355_ZI_BINOMIAL || D == M355_ZI_NEGATIVEBINOMIAL) {
    // zi_*.py rvs: base draw times a Bernoulli(psi) mask
    double u1, u2;
    next_pair(st, u1, u2);
    double base;
    if constexpr (D == M355_ZI_POISSON) base = sample_poisson(st, p[1]);
    else if constexpr (D == M355_ZI_BINOMIAL) base = sample_binomial(st, p[1], p[2]);
    else base = sample_poisson(st, sample_std_gamma(st, p[1]) * ((1.0 - p[2]) / p[2]));
    return (u1 < p[0]) ? base : 0.0;
  } else if constexpr (D == M355_WALD) {  // wald.py:67-68: Michael-Schucany-Haas (numpy's random_wald)
    double u1, u2, z0, z1;
    next_pair(st, u1, u2);
    normal_pair(u1, u2, z0, z1);
    double u3, u4;
    next_pair(st, u3, u4);
    const double mean = p[0], scale = p[1];
    const double y = mean * z0 * z0;
    const double x = mean + mean / (2.0 * scale) * (y - fsqrt(4.0 * scale * y + y * y));
    return (u3 <= mean / (mean + x)) ? x : mean * mean / x;
  } else if constexpr (D == M355_EXGAUSSIAN) {  // exgaussian.py:70-73: normal(mu, sigma) + exponential(scale nu)
    double u1, u2, z0, z1;
    next_pair(st, u1, u2);
    normal_pair(u1, u2, z0, z1);
    double u3, u4;
    next_pair(st, u3, u4);
    return (p[0] + p[1] * z0) + (-flog(u3) * p[2]);
  } else if constexpr (D == M355_POLYAGAMMA) {  // polyagamma.py:191-208: truncated (K = 200) sum-of-gammas representation
    const double z2 = (p[1] * p[1]) / (4.0 * kPi * kPi);
    double acc = 0.0;
    for (int j = 1; j <= 200; ++j) {
      const double d = ((double)j - 0.5) * ((double)j - 0.5) + z2;
      acc += sample_std_gamma(st, p[0]) / d;
    }
    return acc / (2.0 * kPi * kPi);
  } else if constexpr (D == M355_RICE) {  // rice.py:110-113: |(nu + sigma z0, sigma z1)|
    double u1, u2, z0, z1;
    next_pair(st, u1, u2);
    normal_pair(u1, u2, z0, z1);
    const double a = p[0] + p[1] * z0, b = p[1] * z1;
    return fsqrt(a * a + b * b);
  } else if constexpr (D == M355_SKEWNORMAL) {  // skewnormal.py:87-92
    double u1, u2, z0, z1;
    next_pair(st, u1, u2);
    normal_pair(u1, u2, z0, z1);
    const double d = p[2] / fsqrt(1.0 + p[2] * p[2]);
    const double w = d * z0 + z1 * fsqrt(1.0 - d * d);
    const double sg = z0 > 0.0 ? 1.0 : (z0 < 0.0 ? -1.0 : 0.0);
    return sg * w * p[1] + p[0];
  } else if constexpr (D == M355_VONMISES) {  // vonmises.py:49-50: Best & Fisher (numpy's random_vonmises)
    const double mu = p[0], kappa = p[1];
    double u1, u2;
    if (kappa < 1e-8) {
      next_pair(st, u1, u2);
      return kPi * (2.0 * u1 - 1.0);
    }
    const double s = vm_setup(kappa);
    double w = 1.0;
    for (int it = 0; it < kMaxAttempts; ++it) {
      next_pair(st, u1, u2);
      if (vm_attempt(s, kappa, u1, u2, w)) break;
    }
    next_pair(st, u1, u2);
    return vm_finish(w, u1, mu);
  } else if constexpr (D == M355_INVERSEGAMMA) {  // inversegamma.py:69-70: 1 / Gamma(alpha, rate beta)
    return p[1] / sample_std_gamma(st, p[0]);
  } else if constexpr (D == M355_CHISQUARED) {  // chisquared.py:59-60
    return 2.0 * sample_std_gamma(st, 0.5 * p[0]);
  } else if constexpr (D == M355_SCALEDINVERSECHISQUARED) {  // scaledinversechisquared.py:64-65
    return (p[0] * p[1]) / (2.0 * sample_std_gamma(st, 0.5 * p[0]));
  } else if constexpr (D == M355_BETASCALED) {  // betascaled.py:107-109
    if (bb_applies(p[0], p[1])) return sample_beta_bb(st, p[0], p[1]) * (p[3] - p[2]) + p[2];
    const double ga = sample_std_gamma(st, p[0]);
    const double gb = sample_std_gamma(st, p[1]);
    return ga / (ga + gb) * (p[3] - p[2]) + p[2];
  } else if constexpr (D == M355_HALFSTUDENTT) {  // halfstudentt.py:117-119: |sigma * z / sqrt(chi2_nu / nu)|
    // as studentt: the normal is the second Box-Muller output of the accepted gamma attempt's block
    const double a = 0.5 * p[0];
    double z0 = 0.0;
    if (!(a > 0.0) || a == kInf) z0 = normal_single(philox4x32_10(st.idx_lo, st.idx_hi, 0x7fffffffu, st.stream, st.k0, st.k1));
    const double g = sample_std_gamma<true>(st, a, &z0);
    return __builtin_fabs(z0 * fsqrt(a / g) * p[1]);
  } else {
    return kNaN;
  }
}

}  // namespace m355
