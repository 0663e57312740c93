// Multivariate log-densities: batched Cholesky + triangular solve + fused epilogue
// (mvnormal.py:9-37,80-83; mvstudentt.py:58-68; matrixnormal.py:68-94 of the reference).
//
//   k_cholesky   one workgroup per covariance matrix, the matrix in LDS (d <= 128): right-looking
//                factorisation, writes L (lower, row-major, upper part zeroed), 1/diag(L) and
//                logdet = 2*sum(log diag) (mvnormal.py:9-14).  A non-positive pivot gives NaN, as
//                the reference's Cholesky does (posdef is ignored there, quirk Q5).
//   k_solve      one thread per observation: forward substitution L y = x - mu in the order of the
//                reference's per-row solve (mvnormal.py:30-31), y kept in LDS column-major
//                ([component][thread]: conflict-free), L read through wave-uniform (scalar) loads
//                when the covariance is shared; quad = sum y^2 (mvnormal.py:32) and the logpdf
//                epilogue fused.  Generic strides make the same kernel serve both solves of
//                matrixnormal (columns of X-M against L_U, then rows against L_V).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "m355_error.h"
#include "m355_special.h"
#include "../../include/m355_dist.h"

namespace m355 {

constexpr int kMaxDim = 128;
constexpr int kSolveThreads = 64;

// workspace layout per matrix: L (d*d) | invdiag (d) | logdet (1)
__host__ __device__ inline size_t chol_stride(int d) { return (size_t)d * d + d + 1; }

__global__ __launch_bounds__(256) void k_cholesky(const double* __restrict__ cov, int d, double* __restrict__ ws) {
  extern __shared__ double a[];  // d x d, row-major
  const double* src = cov + (size_t)blockIdx.x * d * d;
  double* dst = ws + (size_t)blockIdx.x * chol_stride(d);
  for (int e = threadIdx.x; e < d * d; e += blockDim.x) a[e] = src[e];
  __syncthreads();
  for (int j = 0; j < d; ++j) {
    // pivot and column scaling
    const double piv = sqrt_(a[j * d + j]);  // NaN for a negative pivot
    __syncthreads();
    if (threadIdx.x == 0) a[j * d + j] = piv;
    for (int i = j + 1 + threadIdx.x; i < d; i += blockDim.x) a[i * d + j] = a[i * d + j] / piv;
    __syncthreads();
    // trailing update of the lower triangle: a[i][k] -= a[i][j] * a[k][j], j < k <= i
    const int m = d - j - 1;
    for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
      const int i = j + 1 + e / m, k = j + 1 + e % m;
      if (k <= i) a[i * d + k] = __builtin_fma(-a[i * d + j], a[k * d + j], a[i * d + k]);
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
    const int i = e / d, k = e % d;
    dst[e] = (k <= i) ? a[e] : 0.0;
  }
  for (int i = threadIdx.x; i < d; i += blockDim.x) dst[(size_t)d * d + i] = 1.0 / a[i * d + i];
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += flog(a[i * d + i]);  // log of a NaN / negative diag -> NaN
    dst[(size_t)d * d + d] = 2.0 * s;
  }
}

struct SolveArgs {
  const double* x;       // observation r, component i: x[(r / inner) * s1 + (r % inner) * s2 + i * cs]
  const double* mu;      // same with mu_s1 (0 when mu is shared over the batch)
  long long inner, s1, s2, cs, mu_s1, mu_s2, mu_cs;
  const double* chol;    // chol_stride(d) doubles per matrix
  long long l_div, l_mod;  // matrix of observation r: (r / l_div) % l_mod
  double* y_out;         // optional, laid out like x
  double* out;           // optional, one value per observation
  long long n;
  int d;
  int mode;              // 0: quad, 1: mvnormal logpdf, 2: mvstudentt logpdf
  double nu, c0;         // mvstudentt: nu and the nu-only normaliser; mvnormal: c0 = d*log(2 pi)
};

template <bool SHARED> __global__ __launch_bounds__(kSolveThreads) void k_solve(const SolveArgs a) {
  extern __shared__ double y[];  // [d][kSolveThreads]
  const int t = threadIdx.x;
  const long long r = (long long)blockIdx.x * kSolveThreads + t;
  if (r >= a.n) return;  // no barrier below: threads are independent
  const int d = a.d;
  const long long b = r / a.inner, c = r % a.inner;
  const double* xr = a.x + b * a.s1 + c * a.s2;
  const double* mr = a.mu + b * a.mu_s1 + c * a.mu_s2;
  const long long li = SHARED ? 0 : (r / a.l_div) % a.l_mod;  // SHARED: wave-uniform address -> scalar loads
  const double* L = a.chol + (size_t)li * chol_stride(d);
  const double* invdiag = L + (size_t)d * d;
  double quad = 0.0;
  for (int i = 0; i < d; ++i) {
    // s = delta_i - sum_{k<i} L[i][k] y_k, two interleaved chains to hide the FMA latency
    double s0 = xr[i * a.cs] - mr[i * a.mu_cs], s1 = 0.0;
    const double* Li = L + (size_t)i * d;
    int k = 0;
    for (; k + 1 < i; k += 2) {
      s0 = __builtin_fma(-Li[k], y[k * kSolveThreads + t], s0);
      s1 = __builtin_fma(-Li[k + 1], y[(k + 1) * kSolveThreads + t], s1);
    }
    if (k < i) s0 = __builtin_fma(-Li[k], y[k * kSolveThreads + t], s0);
    const double yi = (s0 + s1) * invdiag[i];
    y[i * kSolveThreads + t] = yi;
    quad = __builtin_fma(yi, yi, quad);
    if (a.y_out) a.y_out[b * a.s1 + c * a.s2 + i * a.cs] = yi;
  }
  if (a.out) {
    const double logdet = L[(size_t)d * d + d];
    double v = quad;
    if (a.mode == 1) {
      v = -0.5 * (a.c0 + logdet + quad);  // mvnormal.py:82-83
    } else if (a.mode == 2) {
      // mvstudentt.py:62-64: norm + inner - 0.5*logdet, norm = c0 (nu- and k-only terms)
      v = a.c0 - 0.5 * (a.nu + (double)d) * flog1p(quad / a.nu) - 0.5 * logdet;
    }
    a.out[r] = v;
  }
}

// out[b] = scale * sum_{j<m} v[b*m + j] + offset[b % noff]   (matrixnormal's Frobenius norm and normaliser)
__global__ __launch_bounds__(256) void k_rowsum(const double* __restrict__ v, long long nb, int m, double scale,
                                                const double* __restrict__ offset, long long noff,
                                                double* __restrict__ out) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  double s = 0.0;
  for (int j = 0; j < m; ++j) s += v[b * m + j];
  out[b] = __builtin_fma(scale, s, offset[b % noff]);
}

// offset[b] = -(0.5*m*n*log(2 pi) + 0.5*n*logdet_U[b % nU] + 0.5*m*logdet_V[b % nV])  (matrixnormal.py:92)
__global__ void k_matn_norm(const double* __restrict__ cholU, long long nU, int m, const double* __restrict__ cholV,
                            long long nV, int n, long long nb, double* __restrict__ offset) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const double ldU = cholU[(size_t)(b % nU) * chol_stride(m) + (size_t)m * m + m];
  const double ldV = cholV[(size_t)(b % nV) * chol_stride(n) + (size_t)n * n + n];
  offset[b] = -(0.5 * m * n * 1.8378770664093454836 + 0.5 * n * ldU + 0.5 * m * ldV);
}

}  // namespace m355

namespace {
using m355::fail;
int launched() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? M355_OK : fail(M355_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
}
int run_cholesky(const double* cov, long long ncov, int d, double* ws, hipStream_t s) {
  const size_t lds = (size_t)d * d * sizeof(double);
  if (lds > 64 * 1024) {  // up to 128 KiB of the CU's 160 KiB LDS for d = 128
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(m355::k_cholesky),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(M355_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(m355::k_cholesky, dim3((unsigned)ncov), dim3(256), (size_t)d * d * sizeof(double), s, cov, d, ws);
  return launched();
}
int run_solve(const m355::SolveArgs& a, bool shared, hipStream_t s) {
  const long long blocks = (a.n + m355::kSolveThreads - 1) / m355::kSolveThreads;
  const size_t lds = (size_t)a.d * m355::kSolveThreads * sizeof(double);
  if (shared) hipLaunchKernelGGL(m355::k_solve<true>, dim3((unsigned)blocks), dim3(m355::kSolveThreads), lds, s, a);
  else hipLaunchKernelGGL(m355::k_solve<false>, dim3((unsigned)blocks), dim3(m355::kSolveThreads), lds, s, a);
  return launched();
}
}  // namespace

extern "C" size_t m355_mvn_workspace_bytes(int d, int64_t ncov) {
  return (size_t)ncov * m355::chol_stride(d) * sizeof(double);
}

extern "C" int m355_mvn_logpdf(int kind, int64_t n, int d, const double* x, const double* mu, int64_t mu_stride,
                               const double* cov, int64_t ncov, int64_t cov_div, int64_t cov_mod, double nu,
                               double* out, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace m355;
  if (kind != 0 && kind != 1) return fail(M355_ERR_INVALID, "kind must be 0 (mvnormal) or 1 (mvstudentt)");
  if (d < 1 || d > kMaxDim) return fail(M355_ERR_UNSUPPORTED, "dimension %d outside 1..%d", d, kMaxDim);
  if (n < 0 || ncov < 1 || cov_div < 1 || cov_mod < 1 || cov_mod > ncov) return fail(M355_ERR_INVALID, "bad n / ncov / cov index");
  if (n == 0) return M355_OK;
  if (!x || !mu || !cov || !out || !workspace) return fail(M355_ERR_INVALID, "NULL pointer argument");
  if (workspace_bytes < m355_mvn_workspace_bytes(d, ncov))
    return fail(M355_ERR_WORKSPACE, "workspace of %zu bytes is smaller than the required %zu", workspace_bytes,
                 m355_mvn_workspace_bytes(d, ncov));
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* ws = static_cast<double*>(workspace);
  int rc = run_cholesky(cov, ncov, d, ws, s);
  if (rc != M355_OK) return rc;
  SolveArgs a{};
  a.x = x;
  a.mu = mu;
  a.inner = 1;
  a.s1 = d;
  a.s2 = 0;
  a.cs = 1;
  a.mu_s1 = mu_stride;
  a.mu_s2 = 0;
  a.mu_cs = 1;
  a.chol = ws;
  a.l_div = cov_div;
  a.l_mod = cov_mod;
  a.y_out = nullptr;
  a.out = out;
  a.n = n;
  a.d = d;
  if (kind == 0) {
    a.mode = 1;
    a.c0 = (double)d * 1.8378770664093454836;  // k*log(2 pi)
  } else {
    a.mode = 2;
    a.nu = nu;
    // gammaln((nu+k)/2) - gammaln(nu/2) - 0.5*k*log(nu*pi)   (mvstudentt.py:62)
    a.c0 = lgamma((nu + d) / 2.0) - lgamma(nu / 2.0) - 0.5 * d * log(nu * 3.14159265358979323846);
  }
  return run_solve(a, ncov == 1, s);
}

extern "C" size_t m355_matrixnormal_workspace_bytes(int64_t nbatch, int m, int n, int64_t nrow, int64_t ncol) {
  return ((size_t)nrow * m355::chol_stride(m) + (size_t)ncol * m355::chol_stride(n) + (size_t)nbatch * m * n +
          (size_t)nbatch * m + (size_t)nbatch) * sizeof(double);
}

// X: nbatch x m x n (row-major, contiguous); mu: m x n shared (mu_batched = 0) or nbatch x m x n;
// rowcov: nrow x m x m, colcov: ncol x n x n with nrow, ncol in {1, nbatch}.
extern "C" int m355_matrixnormal_logpdf(int64_t nbatch, int m, int n, const double* X, const double* mu,
                                        int mu_batched, const double* rowcov, int64_t nrow, const double* colcov,
                                        int64_t ncol, double* out, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  using namespace m355;
  if (m < 1 || n < 1 || m > kMaxDim || n > kMaxDim) return fail(M355_ERR_UNSUPPORTED, "matrix dimensions outside 1..%d", kMaxDim);
  if (nbatch < 0 || (nrow != 1 && nrow != nbatch) || (ncol != 1 && ncol != nbatch)) return fail(M355_ERR_INVALID, "bad batch sizes");
  if (nbatch == 0) return M355_OK;
  if (!X || !mu || !rowcov || !colcov || !out || !workspace) return fail(M355_ERR_INVALID, "NULL pointer argument");
  if (workspace_bytes < m355_matrixnormal_workspace_bytes(nbatch, m, n, nrow, ncol)) return fail(M355_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* wsU = static_cast<double*>(workspace);
  double* wsV = wsU + (size_t)nrow * chol_stride(m);
  double* Y = wsV + (size_t)ncol * chol_stride(n);
  double* rowq = Y + (size_t)nbatch * m * n;
  double* off = rowq + (size_t)nbatch * m;
  int rc = run_cholesky(rowcov, nrow, m, wsU, s);
  if (rc != M355_OK) return rc;
  rc = run_cholesky(colcov, ncol, n, wsV, s);
  if (rc != M355_OK) return rc;
  // Y = L_U^-1 (X - M): the n columns of each m x n matrix are observations of dimension m
  SolveArgs a{};
  a.x = X;
  a.mu = mu;
  a.inner = n;
  a.s1 = (long long)m * n;
  a.s2 = 1;
  a.cs = n;
  a.mu_s1 = mu_batched ? (long long)m * n : 0;
  a.mu_s2 = 1;
  a.mu_cs = n;
  a.chol = wsU;
  a.l_div = n;
  a.l_mod = nrow;
  a.y_out = Y;
  a.out = nullptr;
  a.n = nbatch * n;
  a.d = m;
  a.mode = 0;
  rc = run_solve(a, nrow == 1, s);
  if (rc != M355_OK) return rc;
  // Z = L_V^-1 Y^T: the m rows of Y are observations of dimension n; per-row sum of squares
  (void)hipMemsetAsync(off, 0, sizeof(double), s);  // a zero "mu"
  SolveArgs b{};
  b.x = Y;
  b.mu = off;
  b.inner = m;
  b.s1 = (long long)m * n;
  b.s2 = n;
  b.cs = 1;
  b.mu_s1 = 0;
  b.mu_s2 = 0;
  b.mu_cs = 0;
  b.chol = wsV;
  b.l_div = m;
  b.l_mod = ncol;
  b.y_out = nullptr;
  b.out = rowq;
  b.n = nbatch * m;
  b.d = n;
  b.mode = 0;
  rc = run_solve(b, ncol == 1, s);
  if (rc != M355_OK) return rc;
  // -0.5 * ||Z||_F^2 - log_norm   (matrixnormal.py:90-94); `off` is rewritten after the solve read it
  const unsigned blocks = (unsigned)((nbatch + 255) / 256);
  hipLaunchKernelGGL(k_matn_norm, dim3(blocks), dim3(256), 0, s, wsU, nrow, m, wsV, ncol, n, nbatch, off);
  hipLaunchKernelGGL(k_rowsum, dim3(blocks), dim3(256), 0, s, rowq, nbatch, m, -0.5, off, nbatch, out);
  return launched();
}
