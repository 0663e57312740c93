// Categorical and multinomial (categorical.py:64-112, multinomial.py:35-54 of the reference): the
// integer index / support logic of the path.  Everything that decides an index or a support test
// is done on exact values (int64 observations stay integers; float observations are compared
// before any rounding), so those decisions are bit-exact; the gathered / summed values are fp64.
//
//   k_cat_tables   one workgroup: normalised p = p / sum(p) (categorical.py:7-9), log p and the
//                  sequential cumsum(p) (categorical.py:82,104), plus a flag "cumsum is
//                  non-decreasing" that lets ppf use a binary search (same result as the
//                  reference's masked min whenever the flag holds; linear scan otherwise).
//   k_categorical  one observation per thread per step, tables in LDS (k <= 2048) or global.
//   k_multinomial  one row of k counts per group of G lanes (G = power of two >= k, at most 64; a
//                  whole wavefront strides over longer rows): coalesced row-major reads, four
//                  segmented shuffle reductions (sum lgamma(x+1), sum x log max(p,1e-10), sum x,
//                  any x < 0).
#include <hip/hip_runtime.h>

#include "m355_dists.h"
#include "m355_elemwise.h"
#include "m355_error.h"
#include "m355_rng.h"

namespace m355 {

constexpr int kCatLdsMax = 2048;  // 3 tables x 2048 x 8 B = 48 KiB

__global__ __launch_bounds__(256) void k_cat_tables(const double* __restrict__ p, int k, double* __restrict__ ws) {
  __shared__ double total;
  double* pn = ws;
  double* logp = ws + k;
  double* cs = ws + 2 * (size_t)k;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < k; ++i) s += p[i];
    total = s;
  }
  __syncthreads();
  const double s = total;
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    const double v = p[i] / s;
    pn[i] = v;
    logp[i] = ::log(v);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double c = 0.0;
    bool mono = true;
    for (int i = 0; i < k; ++i) {
      const double prev = c;
      c += pn[i];
      cs[i] = c;
      if (!(c >= prev)) mono = false;
    }
    ws[3 * (size_t)k] = mono ? 1.0 : 0.0;
  }
}

template <bool LDS>
__global__ __launch_bounds__(kBlock) void k_categorical(int fn, const void* __restrict__ x, int xdt,
                                                        const double* __restrict__ ws, int k,
                                                        double* __restrict__ out, long long n) {
  extern __shared__ double tab[];
  const double* logp = ws + k;
  const double* cs = ws + 2 * (size_t)k;
  const bool mono = ws[3 * (size_t)k] != 0.0;
  if (LDS) {
    for (int i = threadIdx.x; i < 2 * k; i += kBlock) tab[i] = ws[k + i];
    __syncthreads();
    logp = tab;
    cs = tab + k;
  }
  const double kd = (double)k;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    double r;
    if (fn == M355_PPF || fn == M355_ISF) {  // categorical.py:100-112
      double q = load1(x, xdt, i);
      if (fn == M355_ISF) q = 1.0 - q;
      int idx = k;
      if (mono) {
        // smallest index with cs[idx] >= q (cs non-decreasing); NaN q: no index, idx = k
        int lo = 0, hi = k;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (cs[mid] >= q) hi = mid; else lo = mid + 1;
        }
        idx = (q == q) ? lo : k;
      } else {
        for (int j = k - 1; j >= 0; --j) idx = (cs[j] >= q) ? j : idx;
      }
      r = ppf_bounds_disc((double)idx, q, 0.0, kd - 1.0);
    } else if (fn == M355_LOGPDF || fn == M355_PDF) {  // categorical.py:64-74, 60-61
      bool valid;
      long long xi;
      if (xdt == M355_I64) {
        xi = static_cast<const long long*>(x)[i];
        valid = xi >= 0 && xi < (long long)k;
      } else {
        const double xv = load1(x, xdt, i);
        valid = (xv >= 0.0) & (xv < kd) & (xv == floor_(xv));
        xi = valid ? (long long)xv : 0;  // the cast of an out-of-range value is never used
      }
      xi = xi < 0 ? 0 : (xi > (long long)k - 1 ? (long long)k - 1 : xi);
      r = valid ? logp[xi] : -kInf;
      if (fn == M355_PDF) r = exp_(r);
    } else {  // cdf family, categorical.py:77-97
      double xv;
      long long xi;
      if (xdt == M355_I64) {
        xi = static_cast<const long long*>(x)[i];
        xv = (double)xi;
      } else {
        xv = load1(x, xdt, i);
        const double xf = floor_(xv);
        const double c = xf < 0.0 ? 0.0 : (xf > kd - 1.0 ? kd - 1.0 : xf);  // NaN -> kd-1 side handled below
        xi = (xv == xv) ? (long long)c : 0;
      }
      xi = xi < 0 ? 0 : (xi > (long long)k - 1 ? (long long)k - 1 : xi);
      double c = cdf_bounds(cs[xi], xv, 0.0, kd - 1.0);
      if (!(xv == xv)) c = kNaN;  // the reference's index of a NaN observation is undefined
      r = c;
      if (fn == M355_LOGCDF) r = flog(c);
      if (fn == M355_SF) r = 1.0 - c;
      if (fn == M355_LOGSF) r = flog1p(-c);
    }
    out[i] = r;
  }
}

// One categorical draw per thread: inverse CDF of one Philox uniform on the normalised cumsum
// (categorical.py:115-117 -> Generator.choice-like semantics; statistically equivalent).
__global__ __launch_bounds__(kBlock) void k_categorical_rvs(const double* __restrict__ ws, int k,
                                                            long long* __restrict__ out, long long n,
                                                            unsigned long long seed, unsigned stream,
                                                            unsigned long long offset) {
  const double* cs = ws + 2 * (size_t)k;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
    const unsigned long long c = offset + (unsigned long long)i;
    const Philox ph = philox4x32_10((unsigned)c, (unsigned)(c >> 32), 0u, stream, (unsigned)seed, (unsigned)(seed >> 32));
    const double u = u53(ph.c[0], ph.c[1]) * cs[k - 1];
    int lo = 0, hi = k - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cs[mid] > u) hi = mid; else lo = mid + 1;
    }
    out[i] = lo;
  }
}

struct MultArgs {
  const void* x;  // rows x k, row-major
  int xdt;
  const double* p;  // k doubles (p_rows == 1) or rows x k
  long long p_rows;
  const double* n_ptr;  // NULL: n_scalar
  double n_scalar;
  long long n_mod;  // n_ptr index = row % n_mod
  double* out;
  long long rows;
  int k;
};

template <int G> __global__ __launch_bounds__(kBlock) void k_multinomial(const MultArgs a) {
  // G lanes per row (G in {1,2,4,...,64}); a group strides over its row when k > G (only G = 64).
  constexpr int kRowsPerBlock = kBlock / G;
  const int g = threadIdx.x / G, l = threadIdx.x % G;
  for (long long row = (long long)blockIdx.x * kRowsPerBlock + g; row < a.rows;
       row += (long long)gridDim.x * kRowsPerBlock) {
    double s_lg = 0.0, s_xlp = 0.0, s_x = 0.0;
    int neg = 0;
    const double* prow = a.p + (a.p_rows == 1 ? 0 : row * a.k);
    for (int j = l; j < a.k; j += G) {
      const double xv = load1(a.x, a.xdt, (unsigned long long)(row * a.k + j));
      const double pv = prow[j];
      s_lg += lgamma1p_obs(xv);
      s_xlp = __builtin_fma(xv, ::log(pv > 1e-10 ? pv : 1e-10), s_xlp);
      s_x += xv;
      neg |= (xv < 0.0) ? 1 : 0;
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
      s_lg += __shfl_xor(s_lg, o, 64);
      s_xlp += __shfl_xor(s_xlp, o, 64);
      s_x += __shfl_xor(s_x, o, 64);
      neg |= __shfl_xor(neg, o, 64);
    }
    if (l == 0) {
      const double nv = a.n_ptr ? a.n_ptr[row % a.n_mod] : a.n_scalar;
      const double res = (lgamma1p_obs(nv) - s_lg) + s_xlp;  // multinomial.py:43-44
      a.out[row] = (neg | (s_x != nv)) ? -kInf : res;        // multinomial.py:45-52
    }
  }
}

// Multinomial draws by conditional binomials (the decomposition NumPy's Generator.multinomial
// uses): x_j ~ Binomial(n - sum_{i<j} x_i, p_j / (1 - sum_{i<j} p_i)), last cell takes the rest.
__global__ __launch_bounds__(kBlock) void k_multinomial_rvs(const double* __restrict__ p, int k, double nn,
                                                            long long* __restrict__ out, long long rows,
                                                            unsigned long long seed, unsigned stream,
                                                            unsigned long long offset) {
  log_table_init();  // sample_binomial uses flog_tab
  for (long long r = (long long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (long long)gridDim.x * kBlock) {
    const unsigned long long c = offset + (unsigned long long)r;
    RngState st;
    st.idx_lo = (unsigned)c;
    st.idx_hi = (unsigned)(c >> 32);
    st.stream = stream;
    st.k0 = (unsigned)seed;
    st.k1 = (unsigned)(seed >> 32);
    st.attempt = 0;
    double remaining = nn, prem = 1.0;
    for (int j = 0; j < k - 1; ++j) {
      double xj = 0.0;
      if (remaining > 0.0) {
        double pj = p[j] / prem;
        pj = pj > 1.0 ? 1.0 : pj;
        xj = sample_binomial(st, remaining, pj);
      }
      out[r * k + j] = (long long)xj;
      remaining -= xj;
      prem -= p[j];
    }
    out[r * k + (k - 1)] = (long long)remaining;
  }
}

}  // namespace m355

namespace {
using m355::fail;
int launched() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? M355_OK : fail(M355_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
}
int grid_for(long long n) {
  const long long want = (n + m355::kBlock - 1) / m355::kBlock;
  return (int)(want < 1 ? 1 : (want > m355::kMaxBlocks ? m355::kMaxBlocks : want));
}
}  // namespace

extern "C" size_t m355_categorical_workspace_bytes(int k) { return (3 * (size_t)(k > 0 ? k : 0) + 1) * sizeof(double); }

extern "C" int m355_categorical(int fn, int64_t n, const void* x, int x_dtype, const double* p, int k, double* out,
                                void* workspace, size_t workspace_bytes, void* stream) {
  using namespace m355;
  if (fn < 0 || fn > M355_ISF) return fail(M355_ERR_INVALID, "unknown function id %d", fn);
  if (k < 1) return fail(M355_ERR_INVALID, "categorical needs k >= 1 probabilities");
  if (n < 0) return fail(M355_ERR_INVALID, "n must be >= 0");
  if (x_dtype != M355_F64 && x_dtype != M355_F32 && x_dtype != M355_I64) return fail(M355_ERR_INVALID, "bad x dtype");
  if (n == 0) return M355_OK;
  if (!x || !p || !out || !workspace) return fail(M355_ERR_INVALID, "NULL pointer argument");
  if (workspace_bytes < m355_categorical_workspace_bytes(k)) return fail(M355_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* ws = static_cast<double*>(workspace);
  hipLaunchKernelGGL(k_cat_tables, dim3(1), dim3(256), 0, s, p, k, ws);
  if (k <= kCatLdsMax)
    hipLaunchKernelGGL(k_categorical<true>, dim3(grid_for(n)), dim3(kBlock), 2 * (size_t)k * sizeof(double), s, fn, x,
                       x_dtype, ws, k, out, (long long)n);
  else
    hipLaunchKernelGGL(k_categorical<false>, dim3(grid_for(n)), dim3(kBlock), 0, s, fn, x, x_dtype, ws, k, out,
                       (long long)n);
  return launched();
}

extern "C" int m355_categorical_rvs(int64_t n, const double* p, int k, uint64_t seed, uint32_t stream_id,
                                    uint64_t counter_offset, int64_t* out, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  using namespace m355;
  if (k < 1 || n < 0) return fail(M355_ERR_INVALID, "bad k / n");
  if (n == 0) return M355_OK;
  if (!p || !out || !workspace) return fail(M355_ERR_INVALID, "NULL pointer argument");
  if (workspace_bytes < m355_categorical_workspace_bytes(k)) return fail(M355_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* ws = static_cast<double*>(workspace);
  hipLaunchKernelGGL(k_cat_tables, dim3(1), dim3(256), 0, s, p, k, ws);
  hipLaunchKernelGGL(k_categorical_rvs, dim3(grid_for(n)), dim3(kBlock), 0, s, ws, k, reinterpret_cast<long long*>(out),
                     (long long)n, (unsigned long long)seed, stream_id, (unsigned long long)counter_offset);
  return launched();
}

extern "C" int m355_multinomial_logpdf(int64_t rows, int k, const void* x, int x_dtype, const double* n_ptr,
                                       double n_scalar, int64_t n_mod, const double* p, int64_t p_rows, double* out,
                                       void* stream) {
  using namespace m355;
  if (k < 1 || rows < 0) return fail(M355_ERR_INVALID, "bad rows / k");
  if (x_dtype != M355_F64 && x_dtype != M355_F32 && x_dtype != M355_I64) return fail(M355_ERR_INVALID, "bad x dtype");
  if (p_rows != 1 && p_rows != rows) return fail(M355_ERR_INVALID, "p must have 1 or `rows` rows");
  if (n_ptr && n_mod < 1) return fail(M355_ERR_INVALID, "n_mod must be >= 1");
  if (rows == 0) return M355_OK;
  if (!x || !p || !out) return fail(M355_ERR_INVALID, "NULL pointer argument");
  MultArgs a{x, x_dtype, p, (long long)p_rows, n_ptr, n_scalar, (long long)(n_ptr ? n_mod : 1), out, (long long)rows, k};
  hipStream_t s = static_cast<hipStream_t>(stream);
  int G = 1;
  while (G < k && G < 64) G <<= 1;
  const long long rpb = kBlock / G;
  const long long want = (rows + rpb - 1) / rpb;
  const int blocks = (int)(want > kMaxBlocks ? kMaxBlocks : want);
  switch (G) {
    case 1: hipLaunchKernelGGL(k_multinomial<1>, dim3(blocks), dim3(kBlock), 0, s, a); break;
    case 2: hipLaunchKernelGGL(k_multinomial<2>, dim3(blocks), dim3(kBlock), 0, s, a); break;
    case 4: hipLaunchKernelGGL(k_multinomial<4>, dim3(blocks), dim3(kBlock), 0, s, a); break;
    case 8: hipLaunchKernelGGL(k_multinomial<8>, dim3(blocks), dim3(kBlock), 0, s, a); break;
    case 16: hipLaunchKernelGGL(k_multinomial<16>, dim3(blocks), dim3(kBlock), 0, s, a); break;
    case 32: hipLaunchKernelGGL(k_multinomial<32>, dim3(blocks), dim3(kBlock), 0, s, a); break;
    default: hipLaunchKernelGGL(k_multinomial<64>, dim3(blocks), dim3(kBlock), 0, s, a); break;
  }
  return launched();
}

extern "C" int m355_multinomial_rvs(int64_t rows, int k, double n, const double* p, uint64_t seed, uint32_t stream_id,
                                    uint64_t counter_offset, int64_t* out, void* stream) {
  using namespace m355;
  if (k < 1 || rows < 0 || !(n >= 0.0)) return fail(M355_ERR_INVALID, "bad rows / k / n");
  if (rows == 0) return M355_OK;
  if (!p || !out) return fail(M355_ERR_INVALID, "NULL pointer argument");
  hipLaunchKernelGGL(k_multinomial_rvs, dim3(grid_for(rows)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), p, k, n,
                     reinterpret_cast<long long*>(out), (long long)rows, (unsigned long long)seed, stream_id,
                     (unsigned long long)counter_offset);
  return launched();
}
