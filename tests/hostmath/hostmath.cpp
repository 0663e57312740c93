// Host (g++) build of the device scalar bodies, for CPU unit tests only: lets the series,
// continued fractions and root finders of csrc/m355_special.h be checked against SciPy while no
// GPU is attached.  The hardware seeds (v_rcp_f64, v_frexp_*) and the device libm (lgamma, exp,
// erfc) are replaced by host equivalents, so this is NOT the parity test: parity is tests -m gpu.
#include <type_traits>
#include <vector>

#include "../../distributions_amd/csrc/m355_dist_list.h"
#include "../../distributions_amd/csrc/m355_dists4.h"
#include "../../distributions_amd/csrc/m355_grad.h"
#include "../../distributions_amd/csrc/m355_moments.h"

using namespace m355;

template <int D, int FN>
static void run(long n, const double* x, const double* p, double* out) {
  const auto k = Dist<D>::derive(p);
  for (long i = 0; i < n; ++i) out[i] = Dist<D>::template eval<FN>(x[i], k);
}

template <int D>
static int by_fn(int fn, long n, const double* x, const double* p, double* out) {
  switch (fn) {
    case M355_LOGPDF: run<D, M355_LOGPDF>(n, x, p, out); return 0;
    case M355_LOGCDF: run<D, M355_LOGCDF>(n, x, p, out); return 0;
    case M355_LOGSF: run<D, M355_LOGSF>(n, x, p, out); return 0;
    case M355_PPF: run<D, M355_PPF>(n, x, p, out); return 0;
    case M355_PDF: run<D, M355_PDF>(n, x, p, out); return 0;
    case M355_CDF: run<D, M355_CDF>(n, x, p, out); return 0;
    case M355_SF: run<D, M355_SF>(n, x, p, out); return 0;
    case M355_ISF: run<D, M355_ISF>(n, x, p, out); return 0;
  }
  return -1;
}

extern "C" int hm_eval(int dist, int fn, long n, const double* x, const double* p, double* out) {
  switch (dist) {
#define HM_CASE(ID, name, np) \
  case ID: return by_fn<ID>(fn, n, x, p, out);
    M355_DIST_LIST(HM_CASE)
#undef HM_CASE
  }
  return -1;
}

// Dist<D>::SumAcc (sum of logpdf with the logarithms carried as products), one "thread" over all n elements,
// renormalised every 256 elements as k_scalar does.  Returns 0 and the sum, or 1 if sum_ok() declines.
template <int D> static int run_sum_acc(long n, const double* x, const double* p, double* out) {
  using DD = Dist<D>;
  const auto k = DD::derive(p);
  if (!DD::sum_ok(k)) return 1;
  typename DD::SumAcc s;
  for (long i = 0; i < n; ++i) {
    DD::sum_add(s, x[i], k);
    if ((i & 255) == 255) DD::sum_renorm(s);
  }
  *out = DD::sum_finish(s, k);
  return 0;
}
template <class T, class = void> struct hm_has_sum_acc : std::false_type {};
template <class T> struct hm_has_sum_acc<T, std::void_t<typename T::SumAcc>> : std::true_type {};
template <int D> static int sum_acc_or_none(long n, const double* x, const double* p, double* out) {
  if constexpr (hm_has_sum_acc<Dist<D>>::value) return run_sum_acc<D>(n, x, p, out);
  else return -2;
}
extern "C" int hm_sum_acc(int dist, long n, const double* x, const double* p, double* out) {
  switch (dist) {
#define HM_SUM(ID, name, np) \
  case ID: return sum_acc_or_none<ID>(n, x, p, out);
    M355_DIST_LIST(HM_SUM)
#undef HM_SUM
  }
  return -1;
}

extern "C" void hm_log(long n, const double* x, double* out) { for (long i = 0; i < n; ++i) out[i] = flog(x[i]); }
extern "C" void hm_log1p(long n, const double* x, double* out) { for (long i = 0; i < n; ++i) out[i] = flog1p(x[i]); }
extern "C" void hm_logphi_tab(long n, const double* x, double* out) { for (long i = 0; i < n; ++i) out[i] = normal_logcdf_z_tab(x[i]); }
extern "C" void hm_erfcx(long n, const double* x, double* out) { for (long i = 0; i < n; ++i) out[i] = erfcx_(x[i]); }
extern "C" void hm_erfinv(long n, const double* x, double* out) { for (long i = 0; i < n; ++i) out[i] = erfinv_(x[i]); }
extern "C" void hm_gammainc(long n, const double* a, const double* x, double* out) {
  for (long i = 0; i < n; ++i) out[i] = gammainc_p(a[i], x[i], lgamma_(a[i]));
}
extern "C" void hm_gammaincc(long n, const double* a, const double* x, double* out) {
  for (long i = 0; i < n; ++i) out[i] = gammainc_q(a[i], x[i], lgamma_(a[i]));
}
extern "C" void hm_gammaincinv(long n, const double* a, const double* p, double* out) {
  for (long i = 0; i < n; ++i) out[i] = gammaincinv_p(a[i], p[i], lgamma_(a[i]));
}
extern "C" void hm_betainc(long n, const double* a, const double* b, const double* x, double* out) {
  for (long i = 0; i < n; ++i)
    out[i] = betainc_(a[i], b[i], x[i], lbeta_(a[i], b[i]));
}
extern "C" void hm_betaincinv(long n, const double* a, const double* b, const double* p, double* out) {
  for (long i = 0; i < n; ++i)
    out[i] = betaincinv_(a[i], b[i], p[i], lbeta_(a[i], b[i]));
}
extern "C" void hm_lbeta(long n, const double* a, const double* b, double* out) { for (long i = 0; i < n; ++i) out[i] = lbeta_(a[i], b[i]); }
extern "C" void hm_log_stream(long n, const double* x, double* out) { for (long i = 0; i < n; ++i) out[i] = flog_stream(x[i]); }
extern "C" void hm_log_tab(long n, const double* x, double* out) { for (long i = 0; i < n; ++i) out[i] = flog_tab(x[i]); }
extern "C" void hm_log1p_stream(long n, const double* x, double* out) { for (long i = 0; i < n; ++i) out[i] = flog1p_stream(x[i]); }

#include "../../distributions_amd/csrc/m355_rng.h"
extern "C" void hm_sincos2pi(long n, const double* u, double* out) {  // out[2i] = sin, out[2i+1] = cos; n counts outputs / 2
  for (long i = 0; i < n; ++i) sincos_2pi(u[i], out[2 * i], out[2 * i + 1]);
}
extern "C" double hm_u53(unsigned hi, unsigned lo) { return u53(hi, lo); }
extern "C" void hm_philox(const unsigned* ctr, const unsigned* key, unsigned* out) {
  const Philox p = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
  for (int i = 0; i < 4; ++i) out[i] = p.c[i];
}
template <int D> static void rvs_loop(long n, const double* p, unsigned long long seed, unsigned stream,
                                      unsigned long long offset, double* out) {
  for (long i = 0; i < n; ++i) out[i] = sample_one<D>(offset + (unsigned long long)i, seed, stream, p);
}
extern "C" int hm_rvs(int dist, long n, const double* p, unsigned long long seed, unsigned stream,
                      unsigned long long offset, double* out) {
  switch (dist) {
#define HM_RVS_CASE(ID, name, np) \
  case ID: rvs_loop<ID>(n, p, seed, stream, offset, out); return 0;
    M355_DIST_LIST(HM_RVS_CASE)
#undef HM_RVS_CASE
  }
  return -1;
}

// Discrete families with scalar parameters: inversion on the tabulated cdf (csrc/m355_rng.h inv_draw, the body of
// k_rvs_inv).  Returns -2 where the family's window does not fit (the GPU library then takes its per-lane samplers).
template <int D> static int rvs_inv_loop(long n, const double* p, unsigned long long seed, unsigned stream,
                                         unsigned long long offset, double* out) {
  using DD = Dist<D>;
  if (!InvFamily<D>::ok(p)) return -2;
  const auto k = DD::derive(p);
  double lo = 0.0;
  int len = 0;
  if (!DD::dtab_range(k, lo, len) || len < 1) return -2;
  std::vector<double> c(len);
  std::vector<unsigned short> g(kInvGuide + 1);
  for (int j = 0; j < len; ++j) c[j] = DD::dtab_cdf(lo + (double)j, k);
  for (int j = 0; j <= kInvGuide; ++j) g[j] = (unsigned short)(j < kInvGuide ? inv_guide(c.data(), len, j) : len - 1);
  for (long i = 0; i < n; ++i) {
    const unsigned long long e = offset + (unsigned long long)i, cc = e >> 1;
    const Philox ph = philox4x32_10((unsigned)cc, (unsigned)(cc >> 32), 0u, stream, (unsigned)seed, (unsigned)(seed >> 32));
    out[i] = (e & 1ull) ? inv_draw<D>(c.data(), g.data(), len, lo, ph.c[2], ph.c[3], k)
                        : inv_draw<D>(c.data(), g.data(), len, lo, ph.c[0], ph.c[1], k);
  }
  return 0;
}
extern "C" int hm_rvs_inv(int dist, long n, const double* p, unsigned long long seed, unsigned stream,
                          unsigned long long offset, double* out) {
  switch (dist) {
#define HM_INV_CASE(ID) \
  case ID: return rvs_inv_loop<ID>(n, p, seed, stream, offset, out);
    M355_INV_LIST(HM_INV_CASE)
#undef HM_INV_CASE
  }
  return -1;
}

// Gradient path (csrc/m355_grad.h): out[i * (2 + NP) + s] = {logpdf, d/dx, d/dparam_0, ...} of element i.
template <int D> static int grad_loop(long n, const double* x, const double* p, double* out) {
  using G = Grad<D>;
  if constexpr (!G::kHas) {
    return -1;
  } else {
    const auto k = grad_pre<G>(p);
    for (long i = 0; i < n; ++i) grad_eval<G>(x[i], k, out + i * (2 + G::NP));
    return 0;
  }
}
extern "C" int hm_grad(int dist, long n, const double* x, const double* p, double* out) {
  switch (dist) {
#define HM_GRAD_CASE(ID, name, np) \
  case ID: return grad_loop<ID>(n, x, p, out);
    M355_DIST_LIST(HM_GRAD_CASE)
#undef HM_GRAD_CASE
  }
  return -1;
}
// Grad<D>::Stats (value and parameter gradients of the summed logpdf from sufficient statistics), one "thread"
// over all n elements; out[0] = sum logpdf, out[2 + j] = sum d/dparam_j (out[1], d/dx, is not on this path).
template <int D> static int run_grad_stats(long n, const double* x, const double* p, double* out) {
  using G = Grad<D>;
  constexpr int S = 2 + G::NP;
  if (!G::stats_ok(p)) return 1;
  const auto k = grad_pre<G>(p);
  typename G::Stats st;
  for (int s = 0; s < S; ++s) out[s] = 0.0;
  for (long i = 0; i < n; ++i) {
    if (!G::stats_add(st, x[i], k)) {
      double r[S];
      grad_eval<G, 0u>(x[i], k, r);
      for (int s = 0; s < S; ++s) out[s] += r[s];
    }
    if ((i & 255) == 255) G::stats_renorm(st);
  }
  double r[S];
  grad_unpack(G::stats_lp(st, k), r, std::make_integer_sequence<int, 1 + G::NP>{});
  for (int s = 0; s < S; ++s) out[s] += r[s];
  return 0;
}
extern "C" int hm_grad_stats(int dist, long n, const double* x, const double* p, double* out) {
  switch (dist) {
    case M355_GAMMA: return run_grad_stats<M355_GAMMA>(n, x, p, out);
    case M355_BETA: return run_grad_stats<M355_BETA>(n, x, p, out);
    case M355_STUDENTT: return run_grad_stats<M355_STUDENTT>(n, x, p, out);
  }
  return -1;
}

// Mom<D>::eval<SerialCtx>(stat, p) for n parameter sets (p: n x 4, row-major)
extern "C" int hm_moment(int dist, int stat, long n, const double* p, double* out) {
  switch (dist) {
#define HM_MOM(ID, name, np) \
  case ID:                   \
    for (long i = 0; i < n; ++i) out[i] = Mom<ID>::eval<SerialCtx>(stat, p + 4 * i); \
    return 0;
    M355_DIST_LIST(HM_MOM)
#undef HM_MOM
  }
  return -1;
}

extern "C" void hm_lgamma1p_obs(long n, const double* x, double* out) { for (long i = 0; i < n; ++i) out[i] = lgamma1p_obs(x[i]); }
extern "C" void hm_digamma(long n, const double* x, double* out) { for (long i = 0; i < n; ++i) out[i] = digamma_(x[i]); }

// k_tab's table build (m355_elemwise.h) restated on the host for the non-quantile functions: the range, the node values
// through tab_x, the 16-point DCT per cell and the validation.  out = {wlo, whi, tail/scale per cell (64)}.
template <class T, class = void> struct hm_has_tab : std::false_type {};
template <class T> struct hm_has_tab<T, std::void_t<decltype(&T::tab_w)>> : std::true_type {};
template <class T, class = void> struct hm_has_node : std::false_type {};
template <class T> struct hm_has_node<T, std::void_t<decltype(T::kTabNode)>> : std::true_type {};
template <class T, class = void> struct hm_has_cap : std::false_type {};
template <class T> struct hm_has_cap<T, std::void_t<decltype(&T::tab_cap)>> : std::true_type {};
template <class T, class = void> struct hm_has_ppf_node : std::false_type {};
template <class T> struct hm_has_ppf_node<T, std::void_t<decltype(&T::tab_ppf_node)>> : std::true_type {};
template <class T, class = void> struct hm_has_ppf_log : std::false_type {};
template <class T> struct hm_has_ppf_log<T, std::void_t<decltype(&T::tab_ppf_log)>> : std::true_type {};
template <int D, int FN> static int tab_build(const double* p, double* out) {
  using DD = Dist<D>;
  if constexpr (!hm_has_tab<DD>::value) {
    return -2;
  } else {
    const auto k = DD::derive(p);
    if (!DD::tab_ok(k)) return 1;
    double rng[2];
    for (int e = 0; e < 2; ++e) {
      double tq = 1e-9, we = kNaN;
      for (int it = 0; it < 4 && !(we - we == 0.0); ++it, tq *= 100.0)
        we = DD::tab_w(DD::template eval<M355_PPF>(e == 0 ? tq : 1.0 - tq, k), k);
      if constexpr (hm_has_cap<DD>::value) {
        if (!(we - we == 0.0)) we = DD::tab_w(DD::tab_cap(e == 1, k), k);
      }
      rng[e] = we;
    }
    out[0] = rng[0];
    out[1] = rng[1];
    const double wlo = rng[0], h = (rng[1] - rng[0]) / 64.0;
    const double pi = 3.14159265358979323846;
    for (int cell = 0; cell < 64; ++cell) {
      double f[16], c[16];
      for (int j = 0; j < 16; ++j) {
        const double w = wlo + h * (cell + 0.5 + 0.5 * __builtin_cos(pi * (2 * j + 1) / 32.0));
        if constexpr (hm_has_node<DD>::value) f[j] = DD::template tab_node<FN>(w, k);
        else f[j] = DD::template eval<FN>(DD::tab_x(w, k), k);
      }
      double sum = 0.0;
      for (int m = 0; m < 16; ++m) {
        double acc = 0.0;
        for (int j = 0; j < 16; ++j) acc += f[j] * __builtin_cos(pi * m * (2 * j + 1) / 32.0);
        c[m] = acc * (m == 0 ? 1.0 / 16.0 : 2.0 / 16.0);
        sum += __builtin_fabs(c[m]);
      }
      const double scale = sum > 1.0 ? sum : 1.0;
      out[2 + cell] = (__builtin_fabs(c[14]) + __builtin_fabs(c[15])) / scale;
    }
    return 0;
  }
}
extern "C" int hm_tab_build(int dist, int fn, const double* p, double* out) {
  switch (dist) {
#define HM_TAB(ID, name, np)                                                \
  case ID:                                                                  \
    if (fn == M355_LOGCDF) return tab_build<ID, M355_LOGCDF>(p, out);       \
    if (fn == M355_LOGSF) return tab_build<ID, M355_LOGSF>(p, out);         \
    if (fn == M355_CDF) return tab_build<ID, M355_CDF>(p, out);             \
    if (fn == M355_SF) return tab_build<ID, M355_SF>(p, out);               \
    return -1;
    M355_DIST_LIST(HM_TAB)
#undef HM_TAB
  }
  return -1;
}

// The table build of k_tab_build (m355_elemwise.h) with the format as arguments -- ncell cells, nn Chebyshev nodes per
// cell, the first nc coefficients kept -- restated on the host, quantiles included: per cell out[2 + cell] = (sum of the
// |dropped or last two| coefficients) / scale, and err[cell] = the largest |interpolant - node function| / scale over
// 33 points of the cell (the truth against which the validation rule is judged).
template <int D, int FN> static double hm_tab_node(double w, const typename Dist<D>::K& k) {
  using DD = Dist<D>;
  constexpr bool QDOM = FN == M355_PPF || FN == M355_ISF;
  if constexpr (QDOM) {
    if constexpr (hm_has_ppf_node<DD>::value) {
      return DD::tab_ppf_node(w, k);
    } else {
      const double e = exp_(-__builtin_fabs(w));
      const double p = m355::fdiv(e, 1.0 + e);
      double f;
      if (w <= 0.0) {
        f = DD::template eval<M355_PPF>(p, k);
      } else {
        f = DD::template eval<M355_PPF>(1.0 - p, k);
        const double lp = flog(p);
        for (int it = 0; it < 2; ++it) {
          const double lsf = DD::template eval<M355_LOGSF>(f, k);
          const double step = (lsf - lp) * exp_(lsf - DD::template eval<M355_LOGPDF>(f, k));
          if (step - step == 0.0) f += step;
          if (__builtin_fabs(step) <= 1e-9 * __builtin_fabs(f)) break;
        }
      }
      if constexpr (hm_has_ppf_log<DD>::value) {
        if (DD::tab_ppf_log(k)) f = flog(f);
      }
      return f;
    }
  } else if constexpr (hm_has_node<DD>::value) {
    return DD::template tab_node<FN>(w, k);
  } else {
    return DD::template eval<FN>(DD::tab_x(w, k), k);
  }
}
template <int D, int FN> static int tab_build2(const double* p, int ncell, int nn, int nc, double* out, double* err) {
  using DD = Dist<D>;
  if constexpr (!hm_has_tab<DD>::value) {
    return -2;
  } else {
    constexpr bool QDOM = FN == M355_PPF || FN == M355_ISF;
    const auto k = DD::derive(p);
    if (!DD::tab_ok(k)) return 1;
    double rng[2];
    for (int e = 0; e < 2; ++e) {
      double tq = 1e-9, we = kNaN;
      if (QDOM) {
        we = e == 0 ? -20.723265835946928 : 20.723265835946928;
      } else {
        for (int it = 0; it < 4 && !(we - we == 0.0); ++it, tq *= 100.0)
          we = DD::tab_w(DD::template eval<M355_PPF>(e == 0 ? tq : 1.0 - tq, k), k);
        if constexpr (hm_has_cap<DD>::value) {
          if (!(we - we == 0.0)) we = DD::tab_w(DD::tab_cap(e == 1, k), k);
        }
      }
      rng[e] = we;
    }
    out[0] = rng[0];
    out[1] = rng[1];
    const double wlo = rng[0], h = (rng[1] - rng[0]) / ncell;
    const double pi = 3.14159265358979323846;
    bool logmode = false;
    if constexpr (QDOM && hm_has_ppf_log<DD>::value) logmode = DD::tab_ppf_log(k);
    for (int cell = 0; cell < ncell; ++cell) {
      double f[32], c[32];
      for (int j = 0; j < nn; ++j) {
        const double w = wlo + h * (cell + 0.5 + 0.5 * __builtin_cos(pi * (2 * j + 1) / (2.0 * nn)));
        f[j] = hm_tab_node<D, FN>(w, k);
      }
      double sum = 0.0;
      for (int m = 0; m < nn; ++m) {
        double acc = 0.0;
        for (int j = 0; j < nn; ++j) acc += f[j] * __builtin_cos(pi * m * (2 * j + 1) / (2.0 * nn));
        c[m] = acc * (m == 0 ? 1.0 / nn : 2.0 / nn);
        sum += __builtin_fabs(c[m]);
      }
      const double scale = ((QDOM && !logmode) || sum > 1.0) ? sum : 1.0;
      double tail = 0.0;
      if (nc < nn) for (int m = nc; m < nn; ++m) tail += __builtin_fabs(c[m]);
      else tail = __builtin_fabs(c[nn - 2]) + __builtin_fabs(c[nn - 1]);
      out[2 + cell] = tail / scale;
      double worst = 0.0;
      for (int s = 0; s <= 32; ++s) {
        const double t = -1.0 + s / 16.0;
        double b0 = 0.0, b1 = 0.0;  // Clenshaw over the kept coefficients
        for (int m = nc - 1; m >= 1; --m) {
          const double b = c[m] + 2.0 * t * b0 - b1;
          b1 = b0;
          b0 = b;
        }
        const double v = c[0] + t * b0 - b1;
        const double w = wlo + h * (cell + 0.5 + 0.5 * t);
        const double e = __builtin_fabs(v - hm_tab_node<D, FN>(w, k)) / scale;
        if (e > worst || e != e) worst = e;
      }
      err[cell] = worst;
    }
    return 0;
  }
}
extern "C" int hm_tab_build2(int dist, int fn, const double* p, int ncell, int nn, int nc, double* out, double* err) {
  if (nn > 32 || nc > nn || nc < 2) return -3;
  switch (dist) {
#define HM_TAB2(ID, name, np)                                                                   \
  case ID:                                                                                      \
    if (fn == M355_LOGCDF) return tab_build2<ID, M355_LOGCDF>(p, ncell, nn, nc, out, err);      \
    if (fn == M355_LOGSF) return tab_build2<ID, M355_LOGSF>(p, ncell, nn, nc, out, err);        \
    if (fn == M355_PPF) return tab_build2<ID, M355_PPF>(p, ncell, nn, nc, out, err);            \
    return -1;
    M355_DIST_LIST(HM_TAB2)
#undef HM_TAB2
  }
  return -1;
}
