/*
 * m355_dist.h -- C-ABI of libm355dist.so, the MI355X (gfx950) evaluation path for
 * pytensor_distributions' per-distribution functions.
 *
 * The reference has no FFI of its own: its "operator interface" for this path is the module-level
 * function API (normal.logpdf(x, mu, sigma) ...), whose arithmetic PyTensor lowers to a fused
 * Elemwise C loop / RandomVariable.perform / Blockwise LAPACK calls.  Each entry point below is what
 * a PyTensor Op.perform (or any other host) binds instead of that loop; the reference lines it
 * replaces are cited per entry point (paths relative to /root/reference/pytensor_distributions).
 *
 * Conventions
 *   - every function returns 0 on success, a negative M355_ERR_* otherwise; the message of the
 *     last failure on the calling thread is m355_last_error().  Nothing throws across the boundary.
 *   - all data pointers are DEVICE pointers on the current HIP device; the caller owns every
 *     buffer; `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only
 *     enqueue work: they never synchronise, and allocate nothing.
 *   - an operand (x or a parameter) is described by m355_operand; element i of the flattened
 *     broadcast result reads operand element (i / div) % mod:
 *         scalar                : ptr == NULL, value in .scalar
 *         full-size array       : div = 1, mod = n
 *         inner-periodic (x of shape (R, C), parameter of shape (C,)) : div = 1, mod = C
 *         outer (parameter of shape (R, 1))                         : div = C, mod = R
 *     Parameter arrays are always float64; x may be float64, float32 or int64.
 */
#ifndef M355_DIST_H
#define M355_DIST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M355_VERSION 100 /* 0.1.0 */

enum m355_status {
  M355_OK = 0,
  M355_ERR_INVALID = -1,     /* bad argument (NULL pointer, n < 0, bad enum, bad operand) */
  M355_ERR_UNSUPPORTED = -2, /* (dist, fn) pair not on the path */
  M355_ERR_LAUNCH = -3,      /* HIP reported an error at launch */
  M355_ERR_WORKSPACE = -4    /* workspace too small */
};

enum m355_dtype { M355_F64 = 0, M355_F32 = 1, M355_I64 = 2 };

/* distribution ids: the reference module each one replaces */
enum m355_dist {
  M355_NORMAL = 0,                    /* normal.py */
  M355_GAMMA = 1,                     /* gamma.py */
  M355_BETA = 2,                      /* beta.py */
  M355_STUDENTT = 3,                  /* studentt.py */
  M355_POISSON = 4,                   /* poisson.py */
  M355_BINOMIAL = 5,                  /* binomial.py */
  M355_EXPONENTIAL = 6,               /* exponential.py */
  M355_LOGNORMAL = 7,                 /* lognormal.py */
  M355_HALFNORMAL = 8,                /* halfnormal.py */
  M355_CAUCHY = 9,                    /* cauchy.py */
  M355_HALFCAUCHY = 10,                /* halfcauchy.py */
  M355_LAPLACE = 11,                   /* laplace.py */
  M355_LOGISTIC = 12,                  /* logistic.py */
  M355_GUMBEL = 13,                    /* gumbel.py */
  M355_WEIBULL = 14,                   /* weibull.py */
  M355_PARETO = 15,                    /* pareto.py */
  M355_UNIFORM = 16,                   /* uniform.py */
  M355_TRIANGULAR = 17,                /* triangular.py */
  M355_KUMARASWAMY = 18,               /* kumaraswamy.py */
  M355_MOYAL = 19,                     /* moyal.py */
  M355_ASYMMETRICLAPLACE = 20,         /* asymmetriclaplace.py */
  M355_LOGLOGISTIC = 21,               /* loglogistic.py */
  M355_LOGITNORMAL = 22,               /* logitnormal.py */
  M355_INVERSEGAMMA = 23,              /* inversegamma.py */
  M355_CHISQUARED = 24,                /* chisquared.py */
  M355_SCALEDINVERSECHISQUARED = 25,   /* scaledinversechisquared.py */
  M355_HALFSTUDENTT = 26,              /* halfstudentt.py */
  M355_BETASCALED = 27,                /* betascaled.py */
  M355_BERNOULLI = 28,                 /* bernoulli.py */
  M355_GEOMETRIC = 29,                 /* geometric.py */
  M355_DISCRETEUNIFORM = 30,           /* discreteuniform.py */
  M355_DISCRETEWEIBULL = 31,           /* discreteweibull.py */
  M355_NEGATIVEBINOMIAL = 32,          /* negativebinomial.py */
  M355_BETABINOMIAL = 33,              /* betabinomial.py */
  M355_HYPERGEOMETRIC = 34,            /* hypergeometric.py */
  M355_ZI_POISSON = 35,                /* zi_poisson.py */
  M355_ZI_BINOMIAL = 36,               /* zi_binomial.py */
  M355_ZI_NEGATIVEBINOMIAL = 37,       /* zi_negativebinomial.py */
  M355_WALD = 38,                      /* wald.py (ppf: find_ppf, optimization.py:10-32) */
  M355_EXGAUSSIAN = 39,                /* exgaussian.py (ppf: find_ppf) */
  M355_TRUNCATEDNORMAL = 40,           /* truncatednormal.py */
  M355_SKEW_STUDENTT = 41,             /* skew_studentt.py */
  M355_SKEWNORMAL = 42,                /* skewnormal.py (cdf through Owen's T; ppf: find_ppf) */
  M355_VONMISES = 43,                  /* vonmises.py (cdf: helper.py:337-372; ppf: find_ppf) */
  M355_RICE = 44,                      /* rice.py (cdf: ncx2 series / Marcum-Q complement, helper.py:449-539) */
  M355_POLYAGAMMA = 45,                /* polyagamma.py (series density :31-68, trapezoid cdf :124-134, Newton ppf :149-184) */
  M355_DIST_COUNT = 46
};

/* function ids: the reference function of that module */
enum m355_fn {
  M355_LOGPDF = 0, /* normal.py:79-80 gamma.py:77-82 beta.py:102-108 studentt.py:108-119
                      poisson.py:70-71 binomial.py:66-75 */
  M355_LOGCDF = 1, /* normal.py:70-76 gamma.py:65-74 beta.py:90-99 studentt.py:100-105
                      poisson.py:74-75 binomial.py:78-79 */
  M355_LOGSF = 2,  /* normal.py:83-84 gamma.py:85-94 beta.py:111-120 studentt.py:122-123
                      poisson.py:78-79 binomial.py:82-83 */
  M355_PPF = 3,    /* normal.py:58-59 gamma.py:48-50 beta.py:78-79 studentt.py:79-89 (+helper.py:29-54) */
  M355_PDF = 4,    /* normal.py:54-55 gamma.py:44-45 beta.py:74-75 studentt.py:75-76 ... */
  M355_CDF = 5,    /* normal.py:46-47 gamma.py:40-41 beta.py:66-67 studentt.py:67-68
                      poisson.py:49-50 binomial.py:45-46 (+helper.py:4-26) */
  M355_SF = 6,     /* normal.py:62-63 gamma.py:53-54 beta.py:82-83 studentt.py:92-93
                      poisson.py:58-59 binomial.py:54-55 (+helper.py:90-91) */
  M355_ISF = 7     /* normal.py:50-51 gamma.py:57-58 beta.py:70-71 studentt.py:71-72 */
};

typedef struct m355_operand {
  const void* ptr; /* device pointer, or NULL for a scalar */
  double scalar;   /* value when ptr == NULL */
  int64_t div;     /* see header comment; ignored for scalars */
  int64_t mod;
  int32_t dtype;   /* m355_dtype of *ptr (parameters: M355_F64 only) */
  int32_t reserved;
} m355_operand;

int m355_version(void);
const char* m355_last_error(void);
/* Number of HIP devices visible to the library (hipGetDeviceCount). */
int m355_device_count(void);

/*
 * out[i] = fn_dist(x[i], params...) for i in [0, n): replaces the fused Elemwise loop PyTensor
 * compiles for <dist>.<fn>(x, *params) (functions cited at enum m355_fn).  `out_dtype` is
 * M355_F64 or M355_F32.  Parameter order is the reference's (normal: mu, sigma; gamma: alpha,
 * beta; beta: alpha, beta; studentt: nu, mu, sigma; poisson: mu; binomial: n, p; the closed-form
 * and re-parameterised families of SURVEY.md 8(f) rank 1 likewise, e.g. weibull: alpha, beta;
 * triangular: lower, c, upper; betascaled: alpha, beta, lower, upper -- at most four parameters; the discrete
 * families of rank 2 -- bernoulli ... zi_negativebinomial -- too, their ppf / isf by find_ppf_discrete,
 * optimization.py:102-129, where the reference uses it).
 */
int m355_elemwise(int dist, int fn, int out_dtype, int64_t n, const m355_operand* x, int nparams,
                  const m355_operand* params, void* out, void* stream);
/*
 * x and out: `out` may be the very buffer of x (same address, same dtype: every element is read before its result is
 * written); any other overlap of the two ranges is undefined.  Parameter arrays must not overlap out.
 *
 * m355_elemwise_ws: the same call with a device workspace (m355_elemwise_workspace_bytes() bytes recommended, 8-byte
 * aligned, NULL / 0 allowed).  From `min_n` elements on (m355_set_tab_min_n below) the library evaluates logcdf / logsf /
 * cdf / sf / ppf / isf of gamma, beta, studentt (and halfstudentt, chisquared, inversegamma, scaledinversechisquared,
 * betascaled) from tables that a first launch builds from the exact bodies (k_tab_build / k_coltab_build, DESIGN.md
 * section 3: cells that do not reach 1e-14 and arguments outside the table take the exact body) -- and the tables live
 * in this workspace: 20 KiB for scalar parameters; for row-periodic parameters (x of shape (R, P), parameters of shape
 * (P,), P a multiple of 16) 7360 bytes per column plus one bit per element (the elements the tables could not answer,
 * which a last launch redoes by the exact body).  Without a workspace, or with one that is
 * too small for the case at hand, the exact kernels serve (same results within the parity bar, slower).  The workspace
 * is scratch: its contents mean nothing after the call, but calls that may run CONCURRENTLY (different streams) need a
 * workspace each; calls on one stream may share one.  m355_elemwise(...) is m355_elemwise_ws(..., NULL, 0, stream).
 */
size_t m355_elemwise_workspace_bytes(void);
/* what a call with n elements needs at most: period = 0 for scalar parameters, else the length P of the row-periodic
 * parameter arrays (stored logpdf / pdf with row-periodic parameters use 128 P bytes of it for the derived constants:
 * k_periodic_fixed streams x without an LDS table) */
size_t m355_elemwise_workspace_bytes_for(int64_t n, int64_t period);
int m355_elemwise_ws(int dist, int fn, int out_dtype, int64_t n, const m355_operand* x, int nparams,
                     const m355_operand* params, void* out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Tuning knob of m355_elemwise_ws, no counterpart in the reference: the element count from which the launch-time tables
 * are used -- scalar parameters: n >= min_n; row-periodic parameters: n >= 8 * min_n (and >= 8192 rows at the default);
 * also the threshold of the discrete families' lookup kernels (k_dtab: exact values over a window of integers,
 * bit-identical results, no workspace; with row-periodic parameters k_dcolg / k_dcol: one window per column in the
 * workspace, quantiles included, 8 * (2 + 2 * 2048) bytes per column, from 8 * min_n elements and 4096 rows).  Returns the previous threshold; min_n < 0 restores the default (2^21),
 * INT64_MAX switches the tables off.  Process-wide (an atomic: safe beside launches on other threads, which see the old
 * or the new value).
 *
 * Environment variables read once per process, for A/B measurements only (none changes results beyond the parity bar):
 *   M355_TAB_MIN_N=<n>       initial value of the threshold above
 *   M355_TAB_BLOCKS=<b>      grid of k_tab (default 1280, clamped to [1, 32768])
 *   M355_NO_BINNING=1        the divergent functions through k_scalar / k_periodic: no LDS sort, no tables
 *   M355_MVN_BLOCKS=<b>      grid cap of k_solve_mfma (shared covariance, d > 64)
 *   M355_MVN_TWO_LAUNCH=1    shared covariance, d <= 64: k_cholesky + k_solve_mfma instead of the fused k_mvn_shared
 */
int64_t m355_set_tab_min_n(int64_t min_n);
int64_t m355_tab_min_n(void); /* the current threshold */

/*
 * *out_sum = sum_i fn_dist(x[i], params...) in float64: replaces `pt.sum(<dist>.logpdf(...))`
 * (no function of its own in the reference) without materialising the
 * elementwise result.  Deterministic: per-thread partial sums -> wavefront shuffle reduction ->
 * one partial per workgroup in `workspace` -> fixed-order final pass.  `workspace` is device
 * memory of at least m355_sumlogp_workspace_bytes() bytes; out_sum is a device double.
 */
size_t m355_sumlogp_workspace_bytes(void);
int m355_sumlogp(int dist, int fn, int64_t n, const m355_operand* x, int nparams,
                 const m355_operand* params, double* out_sum, void* workspace,
                 size_t workspace_bytes, void* stream);

/*
 * m355_sumlogp without a workspace argument and without any allocation.  NO COUNTERPART IN THE REFERENCE: it exists
 * for callers that cannot allocate device memory around the call -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  The per-workgroup partials live in a file-scope `static __device__`
 * array of the library (4096 partials + one launch-time table for sum-logcdf / sum-logsf of the tabulated families =
 * 6664 doubles = 52.1 KiB on every GPU the library runs on, held until the process ends); its per-device address is
 * looked up once (hipGetSymbolAddress, first call on that device, device current).  With scalar parameters: the same
 * kernels, grid and fixed-order final pass as m355_sumlogp, so *out_sum is bit-identical to it.  With row-periodic
 * parameters the array has room for m355_sumlogp's table of derived constants (k_derive_table + k_periodic_prod) only
 * up to 2568 doubles (P x the family's constants); beyond that this entry point keeps k_periodic with its LDS table,
 * on another grid and in another order of accumulation than m355_sumlogp with its 256 KiB workspace -- the same total
 * within 1e-12 * sum |term|, not bit for bit.
 * The call itself performs no hipMalloc / hipFree and never synchronises.
 *   - n above M355_SUMLOGP_STATIC_MAX_N is rejected with M355_ERR_WORKSPACE (use m355_sumlogp, or split the range
 *     and add the totals).
 *   - every call on one GPU shares that GPU's array: calls that overlap in time on one GPU (different streams or
 *     host threads) race on it -- serialise them, or give concurrent callers m355_sumlogp with a workspace each.
 *     Nothing has to start at zero: each call overwrites the partials it later reads.
 */
#define M355_SUMLOGP_STATIC_MAX_N (INT64_C(1) << 32) /* elements per call (32 GiB of float64) */
int64_t m355_sumlogp_static_max_n(void);
int m355_sumlogp_static(int dist, int fn, int64_t n, const m355_operand* x, int nparams,
                        const m355_operand* params, double* out_sum, void* stream);

/*
 * Value and gradient of the summed log-likelihood in one pass.  The reference has no
 * function for this: its logpdf graphs (e.g. normal.py:79-80, gamma.py:77-82, beta.py:102-108,
 * studentt.py:108-119 and the closed-form families) are differentiated by pytensor.grad; this entry point
 * is what the L_op of a custom Op over m355_elemwise / m355_sumlogp calls.
 *   sums (device, 2 + nparams doubles): [sum logpdf, sum d/dx, sum d/dparam_0, ...] over the n elements of
 *        the broadcast of (x, params) -- for scalar parameters sums[2 + j] is d(sum logpdf)/d(param_j).
 *        sums[1] is NaN (not computed) when x is a full contiguous array and dx is NULL: a caller that does not
 *        ask for d/dx of an array x gets the parameter gradients at the cost of the value alone.
 *   dx (device, n elements of out_dtype, or NULL): d logpdf / dx per element.
 *   dparams (host array of nparams device pointers, or NULL; entries may be NULL): d logpdf / d param_j
 *        per element of the broadcast shape (the caller reduces over the axes the parameter was broadcast
 *        along -- what PyTensor's Elemwise L_op does).
 * Partials follow PyTensor's rules for the primitives of the graph (xlogy0, switch as a select of the taken
 * arm: all partials are 0 outside the support).  Defined for the continuous families (ids 0-3, 6-27) and, with
 * respect to the parameters (and formally x), poisson and binomial; m355_grad_supported(dist) tells.
 * `workspace`: as for m355_sumlogp (same size).  Deterministic (fixed-order reduction).
 */
int m355_grad_supported(int dist);
int m355_logpdf_grad(int dist, int out_dtype, int64_t n, const m355_operand* x, int nparams,
                     const m355_operand* params, double* sums, void* dx, void* const* dparams,
                     void* workspace, size_t workspace_bytes, void* stream);

/*
 * out[i] = one draw of `dist` with the parameters of element i, i in [0, n): replaces
 * pt.random.{normal,gamma,beta,t,poisson,binomial}(..., rng, size) -> numpy.random.Generator
 * (normal.py:66-67, gamma.py:61-62 [shape alpha, scale 1/beta], beta.py:86-87, studentt.py:96-97,
 * poisson.py:66-67, binomial.py:62-63).  Philox4x32-10, key = seed, counter = (counter_offset + i,
 * attempt, stream_id): the draw of global element counter_offset + i does not depend on how the
 * range is split over calls or GPUs.  out_dtype: M355_F64 / M355_F32, or M355_I64 for the discrete
 * distributions (the reference returns int64).  Parameters as in m355_elemwise (float64 arrays
 * or scalars, indexed by (i / div) % mod).  Statistically equivalent to the reference sampler
 * (KS / chi-square), not stream-equal to NumPy's PCG64.
 * The discrete families whose cdf window fits 2048 integers (poisson, binomial, negativebinomial, betabinomial,
 * hypergeometric, zi_*) are sampled by inversion on a tabulated cdf when EVERY parameter is a scalar operand, and by the
 * per-lane rejection / mixture samplers when a parameter is an array: the two forms give different (equally distributed)
 * streams for the same parameter values; within one form the split-invariance above holds.
 * The map (seed, stream, element) -> draw belongs to a build of the library, not to the interface: samplers have been
 * replaced between releases (ziggurat normal, Cheng's beta, the half Student t), so stored draws are reproducible only
 * with the build that made them.
 */
int m355_rvs(int dist, int out_dtype, int64_t n, int nparams, const m355_operand* params, uint64_t seed,
             uint32_t stream_id, uint64_t counter_offset, void* out, void* stream);

/*
 * Moment functions of a distribution, one value per parameter set: out[i] = <dist>.<stat>(*params_i), i in [0, n).
 * Replaces the module-level mean / mode / median / var / std / skewness / kurtosis / entropy of every module of
 * the reference (normal.py:6-43, gamma.py:8-37, beta.py:9-63, ... one block per module) and the helpers they
 * call: continuous_entropy / discrete_entropy / discrete_moment / continuous_moment (+ _skewness, _kurtosis),
 * continuous_mode and zi_mode (helper.py:99-322, 375-446), the Gauss-Hermite sums of logitnormal.py:12-92.
 * Parameters as in m355_elemwise (float64 arrays or scalars, indexed by (i / div) % mod); the helpers that take
 * arange(min, max) or linspace(min, max) of the parameters are evaluated per parameter set.  out_dtype:
 * M355_F64 / M355_F32.  Closed forms run one thread per parameter set; the stats that sum over many points
 * (1000-point trapezoids, sums over a discrete support) run one 64-lane wavefront per parameter set.
 */
enum m355_stat {
  M355_MEAN = 0,
  M355_MODE = 1,
  M355_MEDIAN = 2,
  M355_VAR = 3,
  M355_STD = 4,
  M355_SKEWNESS = 5,
  M355_KURTOSIS = 6,
  M355_ENTROPY = 7,
  M355_STAT_COUNT = 8
};
int m355_moment(int dist, int stat, int out_dtype, int64_t n, int nparams, const m355_operand* params, void* out,
                void* stream);

/*
 * Multivariate normal / Student t log-density: out[r] = logpdf(x[r, :]) for r in [0, n), x row-major
 * n x d (d <= 128).  Replaces mvnormal.logpdf (mvnormal.py:80-83 with quaddist_chol :17-37 and
 * _logdet_from_cholesky :9-14) for kind 0 and mvstudentt.logpdf (mvstudentt.py:58-68) for kind 1:
 * Cholesky of every covariance, one forward substitution per observation row, sum of squares and the epilogue
 * fused (shared covariance with d <= 64: factorisation and solve in one launch, k_mvn_shared).  mu: d doubles shared by all rows
 * (mu_stride 0) or one row per observation (mu_stride d).  cov: ncov matrices d x d; observation
 * r uses matrix (r / cov_div) % cov_mod (shared covariance: ncov = 1).  A covariance that is not
 * positive definite yields NaN, as the reference does (posdef is ignored, mvnormal.py:81).
 * workspace: m355_mvn_workspace_bytes(d, ncov) bytes of device memory (the factors).
 */
size_t m355_mvn_workspace_bytes(int d, int64_t ncov);
int m355_mvn_logpdf(int kind, int64_t n, int d, const double* x, const double* mu, int64_t mu_stride,
                    const double* cov, int64_t ncov, int64_t cov_div, int64_t cov_mod, double nu,
                    double* out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Matrix normal log-density (matrixnormal.py:68-94): X is nbatch x m x n row-major, mu m x n
 * (mu_batched 0) or nbatch x m x n, rowcov nrow x m x m and colcov ncol x n x n with nrow, ncol in
 * {1, nbatch}; out has nbatch entries:  -0.5 * || L_V^-1 (L_U^-1 (X - M))^T ||_F^2 - log_norm.
 */
size_t m355_matrixnormal_workspace_bytes(int64_t nbatch, int m, int n, int64_t nrow, int64_t ncol);
int m355_matrixnormal_logpdf(int64_t nbatch, int m, int n, const double* X, const double* mu,
                             int mu_batched, const double* rowcov, int64_t nrow, const double* colcov,
                             int64_t ncol, double* out, void* workspace, size_t workspace_bytes,
                             void* stream);

/*
 * Categorical (categorical.py:60-112): out[i] = fn(x[i]; p) for one probability vector p of k
 * entries (normalised on the device as p / sum(p), categorical.py:7-9).  fn is an m355_fn:
 * LOGPDF/PDF (:64-74, :60-61: valid iff 0 <= x < k and x == floor(x), gather log p at the clipped
 * integer, else -inf), CDF/LOGCDF/SF/LOGSF (:77-97: gather of the sequential cumsum at
 * clip(floor(x), 0, k-1) through cdf_bounds(., x, 0, k-1)), PPF/ISF (:100-112: smallest index whose
 * cumsum >= q, through ppf_bounds_disc(., q, 0, k-1)).  x may be float64 / float32 / int64 (int64
 * observations are compared as integers); out is float64.  A NaN observation gives -inf (logpdf)
 * or NaN (cdf family, where the reference's index is undefined).
 * workspace: m355_categorical_workspace_bytes(k) bytes of device memory (the three tables).
 */
size_t m355_categorical_workspace_bytes(int k);
int m355_categorical(int fn, int64_t n, const void* x, int x_dtype, const double* p, int k, double* out,
                     void* workspace, size_t workspace_bytes, void* stream);
/* Categorical draws (categorical.py:115-117 -> pt.random.categorical): inverse CDF of one Philox
 * uniform per element, counter = counter_offset + i; int64 out. */
int m355_categorical_rvs(int64_t n, const double* p, int k, uint64_t seed, uint32_t stream_id,
                         uint64_t counter_offset, int64_t* out, void* workspace, size_t workspace_bytes,
                         void* stream);

/*
 * Multinomial log-pmf (multinomial.py:39-54): x is rows x k row-major (float64 / float32 / int64),
 * out[r] = gammaln(n+1) - sum_j gammaln(x[r,j]+1) + sum_j x[r,j] * log(max(p[j], 1e-10)), or -inf
 * when any x[r,j] < 0 or sum_j x[r,j] != n (both decided exactly: integer-valued counts sum exactly
 * in float64 below 2^53).  n: scalar (n_ptr NULL) or array indexed by r % n_mod; p: one row of k
 * (p_rows 1) or rows x k.
 */
int m355_multinomial_logpdf(int64_t rows, int k, const void* x, int x_dtype, const double* n_ptr, double n_scalar,
                            int64_t n_mod, const double* p, int64_t p_rows, double* out, void* stream);
/* Multinomial draws (multinomial.py:57-58 -> pt.random.multinomial): conditional binomials per
 * row, rows x k int64 out; row r uses Philox counter counter_offset + r. */
int m355_multinomial_rvs(int64_t rows, int k, double n, const double* p, uint64_t seed, uint32_t stream_id,
                         uint64_t counter_offset, int64_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* M355_DIST_H */
