// extern "C" boundary of libm355dist.so (declarations and reference citations: include/m355_dist.h).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>

#include "m355_elemwise.h"
#include "m355_error.h"

namespace {
thread_local char g_err[512] = "";
}

namespace m355 {
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace m355

namespace {
using m355::fail;

// Fixed-order sum of the per-workgroup partials (one workgroup): deterministic total.
__global__ __launch_bounds__(m355::kBlock) void k_final(const double* __restrict__ partials, int nparts,
                                                          double* __restrict__ out) {
  // all loads issued before the first add: one memory round trip instead of nparts / 256 of them
  constexpr int kPer = m355::kMaxReduceBlocks / m355::kBlock;
  double v[kPer];
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int i = threadIdx.x + j * m355::kBlock;
    v[j] = i < nparts ? partials[i] : 0.0;
  }
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < kPer; ++j) acc += v[j];
  for (int i = threadIdx.x + kPer * m355::kBlock; i < nparts; i += m355::kBlock) acc += partials[i];
  acc = m355::block_reduce_sum(acc);
  if (threadIdx.x == 0) out[0] = acc;
}

bool convert(const m355_operand& in, long long n, bool is_param, m355::DevOperand& out, const char* what) {
  out.ptr = in.ptr;
  out.scalar = in.scalar;
  out.dtype = in.dtype;
  out.div = 1;
  out.mod = 1;
  if (in.ptr == nullptr) return true;
  if (in.dtype != M355_F64 && (is_param || (in.dtype != M355_F32 && in.dtype != M355_I64))) {
    fail(M355_ERR_INVALID, "%s: dtype %d not accepted here", what, in.dtype);
    return false;
  }
  if (in.div < 1 || in.mod < 1) {
    fail(M355_ERR_INVALID, "%s: div and mod must be >= 1 (got %lld, %lld)", what, (long long)in.div,
         (long long)in.mod);
    return false;
  }
  out.div = (unsigned long long)in.div;
  out.mod = (unsigned long long)in.mod;
  // i / div == 0 for every i < n: the operand is its element 0.  Normalised here so that the kernels' 32-bit index
  // path (n < 2^32) never sees a divisor it would truncate.
  if (n > 0 && in.div >= n) {
    out.div = (unsigned long long)n;
    out.mod = 1;
  }
  return true;
}

int nparams_of(int dist) {
  switch (dist) {
#define M355_NP_CASE(ID, name, np) \
  case ID: return np;
    M355_DIST_LIST(M355_NP_CASE)
#undef M355_NP_CASE
  }
  return -1;
}

int build_args(int dist, int fn, int out_dtype, int64_t n, const m355_operand* x, int nparams,
               const m355_operand* params, m355::LaunchArgs& a) {
  const int np = nparams_of(dist);
  if (np < 0) return fail(M355_ERR_INVALID, "unknown distribution id %d", dist);
  if (!m355::supported(dist, fn)) return fail(M355_ERR_UNSUPPORTED, "function %d of distribution %d is not on the path", fn, dist);
  if (nparams != np) return fail(M355_ERR_INVALID, "distribution %d takes %d parameters, got %d", dist, np, nparams);
  if (n < 0) return fail(M355_ERR_INVALID, "n must be >= 0");
  if (x == nullptr || params == nullptr) return fail(M355_ERR_INVALID, "x and params must not be NULL");
  if (out_dtype != M355_F64 && out_dtype != M355_F32) return fail(M355_ERR_INVALID, "out_dtype must be f64 or f32");
  a.fn = fn;
  a.out_dtype = out_dtype;
  a.n = n;
  a.nparams = np;
  if (!convert(*x, n, false, a.x, "x")) return M355_ERR_INVALID;
  for (int q = 0; q < m355::kMaxParams; ++q) {
    a.p[q] = m355::DevOperand{nullptr, 0.0, 1, 1, M355_F64};
    if (q < np && !convert(params[q], n, true, a.p[q], "parameter")) return M355_ERR_INVALID;
  }
  return M355_OK;
}

int dispatch(int dist, const m355::LaunchArgs& a) {
  switch (dist) {
#define M355_LAUNCH_CASE(ID, name, np) \
  case ID: return m355_launch_##name(a);
    M355_DIST_LIST(M355_LAUNCH_CASE)
#undef M355_LAUNCH_CASE
  }
  return -1000;
}

int check_launch(int rc) {
  if (rc == 0) return M355_OK;
  if (rc == -1000) return fail(M355_ERR_UNSUPPORTED, "(dist, fn) pair has no kernel");
  return fail(M355_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString((hipError_t)rc));
}

}  // namespace

extern "C" int m355_rvs_impl(int dist, int out_dtype, long long n, const m355::GenericOps* ops, unsigned long long seed,
                             unsigned stream_id, unsigned long long offset, void* out, hipStream_t s);

extern "C" int m355_grad_impl(int dist, long long n, const m355::GenericOps* ops, int nparams, void* dx,
                              void* const* dparams, int out_dtype, double* sums, double* partials,
                              hipStream_t stream);

extern "C" {

int m355_version(void) { return M355_VERSION; }

const char* m355_last_error(void) { return g_err; }

int m355_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// covers config 2's row-periodic layout (P = 1024, n = 1e8: 9.2 MiB of tables + 11.9 MiB of miss words)
size_t m355_elemwise_workspace_bytes(void) { return (size_t)32 << 20; }
size_t m355_elemwise_workspace_bytes_for(int64_t n, int64_t period) {
  if (n <= 0 || period < 0) return 0;
  if (period == 0) return (size_t)m355::kTabDoubles * sizeof(double);
  // the derived constants of the P parameter sets (k_periodic_fixed: stored logpdf / pdf), and, for periods the table
  // kernels take, the per-column tables and miss words of logcdf / logsf / cdf / sf / ppf / isf
  size_t need = (size_t)period * 16 * sizeof(double);
  {  // the discrete families' windows (k_dcol): two header doubles and up to 2 x kDTabMax values per column
    const size_t t = ((size_t)period * (2 + 2 * (size_t)m355::kDTabMax) + m355::kDColHdr) * sizeof(double);  // (quantiles: cdf and pmf)
    need = t > need ? t : need;
  }
  if (period % m355::kCtCols == 0) {
    const size_t t = m355::coltab_ws_doubles<M355_LOGCDF>((unsigned long long)period, (long long)(n / period)) * sizeof(double);
    need = t > need ? t : need;
  }
  return need;
}

int m355_elemwise_ws(int dist, int fn, int out_dtype, int64_t n, const m355_operand* x, int nparams,
                     const m355_operand* params, void* out, void* workspace, size_t workspace_bytes, void* stream) {
  m355::LaunchArgs a{};
  const int rc = build_args(dist, fn, out_dtype, n, x, nparams, params, a);
  if (rc != M355_OK) return rc;
  if (n == 0) return M355_OK;
  if (out == nullptr) return fail(M355_ERR_INVALID, "out must not be NULL");
  if (workspace != nullptr && (reinterpret_cast<uintptr_t>(workspace) & 7) != 0)
    return fail(M355_ERR_INVALID, "workspace must be 8-byte aligned");
  a.out = out;
  a.partials = nullptr;
  a.nblocks_out = nullptr;
  a.ews = static_cast<double*>(workspace);
  a.ews_doubles = workspace != nullptr ? workspace_bytes / sizeof(double) : 0;
  a.stream = static_cast<hipStream_t>(stream);
  return check_launch(dispatch(dist, a));
}

int m355_elemwise(int dist, int fn, int out_dtype, int64_t n, const m355_operand* x, int nparams,
                  const m355_operand* params, void* out, void* stream) {
  return m355_elemwise_ws(dist, fn, out_dtype, n, x, nparams, params, out, nullptr, 0, stream);
}

// What callers are told to bring (kMaxBlocks doubles: kept from the first release so that existing workspaces stay
// valid); what the kernels can touch is kMaxReduceBlocks partials (launch_fn caps every REDUCE grid there).
size_t m355_sumlogp_workspace_bytes(void) { return (size_t)m355::kMaxBlocks * sizeof(double); }
static size_t sumlogp_workspace_min_bytes() { return (size_t)m355::kMaxReduceBlocks * sizeof(double); }

int m355_sumlogp(int dist, int fn, int64_t n, const m355_operand* x, int nparams,
                 const m355_operand* params, double* out_sum, void* workspace, size_t workspace_bytes,
                 void* stream) {
  m355::LaunchArgs a{};
  const int rc = build_args(dist, fn, M355_F64, n, x, nparams, params, a);
  if (rc != M355_OK) return rc;
  if (fn != M355_LOGPDF && fn != M355_LOGCDF && fn != M355_LOGSF)
    return fail(M355_ERR_UNSUPPORTED, "sum-logp is defined for logpdf/logcdf/logsf only");
  if (out_sum == nullptr || workspace == nullptr) return fail(M355_ERR_INVALID, "out_sum and workspace must not be NULL");
  if (workspace_bytes < sumlogp_workspace_min_bytes())
    return fail(M355_ERR_WORKSPACE, "workspace of %zu bytes is smaller than the %zu the kernels need (%zu recommended: "
                "m355_sumlogp_workspace_bytes())", workspace_bytes, sumlogp_workspace_min_bytes(),
                m355_sumlogp_workspace_bytes());
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n == 0) {
    hipError_t e = hipMemsetAsync(out_sum, 0, sizeof(double), s);
    return e == hipSuccess ? M355_OK : fail(M355_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
  }
  int nblocks = 0;
  a.out = nullptr;
  a.partials = static_cast<double*>(workspace);
  a.nblocks_out = &nblocks;
  a.stream = s;
  // what the workspace holds beyond the partials serves as scratch for a table of derived constants (row-periodic
  // parameters, k_periodic_prod); m355_sumlogp_static's array has room for one launch-time table (kTabDoubles) and keeps the LDS-table kernel for
  // larger derived-constant tables
  if (workspace_bytes > sumlogp_workspace_min_bytes()) {
    a.table = static_cast<double*>(workspace) + m355::kMaxReduceBlocks;
    a.table_doubles = (long long)((workspace_bytes - sumlogp_workspace_min_bytes()) / sizeof(double));
  }
  const int rc2 = check_launch(dispatch(dist, a));
  if (rc2 != M355_OK) return rc2;
  hipLaunchKernelGGL(k_final, dim3(1), dim3(m355::kBlock), 0, s, static_cast<const double*>(workspace), nblocks,
                     out_sum);
  return check_launch((int)hipGetLastError());
}

// ---- m355_sumlogp_static: the sum-logp path for callers that cannot allocate (include/m355_dist.h) ----------------
// The per-workgroup partials live in this file-scope device array instead of a caller workspace.  `scratch` names the
// array for device code only: host code uses the per-device address obtained once from hipGetSymbolAddress.
}  // extern "C"
namespace {
// 4096 partials (32 KiB) + one launch-time table (sum_logcdf / sum_logsf of the tabulated families: kTabDoubles doubles,
// 20.1 KiB): 52.1 KiB per GPU, held until process exit
constexpr int kStaticDoubles = m355::kMaxReduceBlocks + m355::kTabDoubles;
static __device__ __align__(256) double scratch[kStaticDoubles];
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
double* g_scratch_ptr[kMaxDevices] = {};

// Address of `scratch` on the current device (looked up once per device, with that device current).
int static_scratch(double** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(M355_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= kMaxDevices) return fail(M355_ERR_INVALID, "device ordinal %d beyond the %d supported", dev, kMaxDevices);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[dev] == nullptr) {
    double* p = nullptr;
    e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr)
      return fail(M355_ERR_LAUNCH, "hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return M355_OK;
}
}  // namespace
extern "C" {

int64_t m355_sumlogp_static_max_n(void) { return M355_SUMLOGP_STATIC_MAX_N; }

int64_t m355_tab_min_n(void) { return m355::tab_min_n_ref().load(); }

int64_t m355_set_tab_min_n(int64_t min_n) {
  return m355::tab_min_n_ref().exchange(min_n < 0 ? m355::kTabMinN : (long long)min_n);
}

int m355_sumlogp_static(int dist, int fn, int64_t n, const m355_operand* x, int nparams, const m355_operand* params,
                        double* out_sum, void* stream) {
  if (n > M355_SUMLOGP_STATIC_MAX_N)
    return fail(M355_ERR_WORKSPACE, "m355_sumlogp_static: n = %lld exceeds the maximum of %lld elements per call",
                (long long)n, (long long)M355_SUMLOGP_STATIC_MAX_N);
  double* ws = nullptr;
  const int rc = static_scratch(&ws);
  if (rc != M355_OK) return rc;
  // same kernels, same grid and same fixed-order final pass as m355_sumlogp: the totals are bit-identical.  Every
  // launch writes partials [0, nblocks) before k_final reads exactly those, so nothing has to start at zero.
  return m355_sumlogp(dist, fn, n, x, nparams, params, out_sum, ws, sizeof(double) * kStaticDoubles, stream);
}

int m355_logpdf_grad(int dist, int out_dtype, int64_t n, const m355_operand* x, int nparams,
                     const m355_operand* params, double* sums, void* dx, void* const* dparams, void* workspace,
                     size_t workspace_bytes, void* stream) {
  m355::LaunchArgs a{};
  const int rc = build_args(dist, M355_LOGPDF, out_dtype, n, x, nparams, params, a);
  if (rc != M355_OK) return rc;
  if (!m355_grad_supported(dist)) return fail(M355_ERR_UNSUPPORTED, "distribution %d has no gradient kernel", dist);
  if (sums == nullptr || workspace == nullptr) return fail(M355_ERR_INVALID, "sums and workspace must not be NULL");
  if (workspace_bytes < m355_sumlogp_workspace_bytes())
    return fail(M355_ERR_WORKSPACE, "workspace of %zu bytes is smaller than the required %zu", workspace_bytes,
                m355_sumlogp_workspace_bytes());
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n == 0) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * (2 + nparams), s);
    return e == hipSuccess ? M355_OK : fail(M355_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
  }
  m355::GenericOps ops;
  ops.x = a.x;
  for (int q = 0; q < m355::kMaxParams; ++q) ops.p[q] = a.p[q];
  return check_launch(m355_grad_impl(dist, n, &ops, nparams, dx, dparams, out_dtype, sums, static_cast<double*>(workspace), s));
}

int m355_rvs(int dist, int out_dtype, int64_t n, int nparams, const m355_operand* params, uint64_t seed,
             uint32_t stream_id, uint64_t counter_offset, void* out, void* stream) {
  const int np = nparams_of(dist);
  if (np < 0) return fail(M355_ERR_INVALID, "unknown distribution id %d", dist);
  if (nparams != np) return fail(M355_ERR_INVALID, "distribution %d takes %d parameters, got %d", dist, np, nparams);
  if (n < 0) return fail(M355_ERR_INVALID, "n must be >= 0");
  if (params == nullptr) return fail(M355_ERR_INVALID, "params must not be NULL");
  const bool discrete = dist == M355_POISSON || dist == M355_BINOMIAL ||
                        (dist >= M355_BERNOULLI && dist <= M355_ZI_NEGATIVEBINOMIAL);
  if (out_dtype != M355_F64 && out_dtype != M355_F32 && !(discrete && out_dtype == M355_I64))
    return fail(M355_ERR_INVALID, "out_dtype %d not valid for distribution %d", out_dtype, dist);
  if (n == 0) return M355_OK;
  if (out == nullptr) return fail(M355_ERR_INVALID, "out must not be NULL");
  m355::GenericOps ops;
  ops.x = m355::DevOperand{nullptr, 0.0, 1, 1, M355_F64};
  for (int q = 0; q < m355::kMaxParams; ++q) {
    ops.p[q] = m355::DevOperand{nullptr, 0.0, 1, 1, M355_F64};
    if (q < np && !convert(params[q], n, true, ops.p[q], "parameter")) return M355_ERR_INVALID;
  }
  return check_launch(m355_rvs_impl(dist, out_dtype, n, &ops, seed, stream_id, counter_offset, out,
                                    static_cast<hipStream_t>(stream)));
}

int m355_moment(int dist, int stat, int out_dtype, int64_t n, int nparams, const m355_operand* params, void* out,
                void* stream) {
  const int np = nparams_of(dist);
  if (np < 0) return fail(M355_ERR_INVALID, "unknown distribution id %d", dist);
  if (stat < 0 || stat >= M355_STAT_COUNT) return fail(M355_ERR_INVALID, "unknown moment function id %d", stat);
  if (nparams != np) return fail(M355_ERR_INVALID, "distribution %d takes %d parameters, got %d", dist, np, nparams);
  if (n < 0) return fail(M355_ERR_INVALID, "n must be >= 0");
  if (params == nullptr) return fail(M355_ERR_INVALID, "params must not be NULL");
  if (out_dtype != M355_F64 && out_dtype != M355_F32) return fail(M355_ERR_INVALID, "out_dtype %d not valid for a moment", out_dtype);
  if (n == 0) return M355_OK;
  if (out == nullptr) return fail(M355_ERR_INVALID, "out must not be NULL");
  m355::MomentArgs a;
  a.stat = stat;
  a.out_dtype = out_dtype;
  a.n = n;
  a.out = out;
  a.stream = static_cast<hipStream_t>(stream);
  a.ops.x = m355::DevOperand{nullptr, 0.0, 1, 1, M355_F64};
  for (int q = 0; q < m355::kMaxParams; ++q) {
    a.ops.p[q] = m355::DevOperand{nullptr, 0.0, 1, 1, M355_F64};
    if (q < np && !convert(params[q], n, true, a.ops.p[q], "parameter")) return M355_ERR_INVALID;
  }
  switch (dist) {
#define M355_MOMENT_CASE(ID, name, np_) \
  case ID: return check_launch(m355_launch_moment_##name(a));
    M355_DIST_LIST(M355_MOMENT_CASE)
#undef M355_MOMENT_CASE
  }
  return fail(M355_ERR_INVALID, "unknown distribution id %d", dist);
}

}  // extern "C"
