"""CPU oracle: NumPy/SciPy float64 restatement of the reference's hot-path formulas.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
The product path (``distributions_amd``) never imports, calls or falls back to anything
in this directory; it fails loudly when the HIP library or a GPU is missing.

What it restates (file:line refer to ``/root/reference/pytensor_distributions``):
the reference builds symbolic PyTensor graphs whose arithmetic lives in the un-vendored
third-party dependency ``pytensor>=2.32.0`` (pyproject.toml:24, no lock file), which is not
installed here and cannot be fetched.  Each function below evaluates the same
expression eagerly with ``pt.X -> scipy.special.X`` and ``pt.switch -> np.where`` (a *select*:
both arms are evaluated, NaNs of the dead arm never leak).

How it is pinned: the reference's own tests compare every function against ``scipy.stats`` at
rtol 1e-6 on fixed parameter grids (tests/helper_scipy.py:8-240, tests/test_normal.py:12-18,
tests/test_gamma.py:12-19, tests/test_beta.py:12-17, tests/test_studentt.py:12-16,
tests/test_poisson.py:12-16, tests/test_binomial.py:13-23, tests/test_mvnormal.py:12-37,
tests/test_mvstudentt.py, tests/test_matrixnormal.py, tests/test_categorical.py,
tests/test_multinomial.py).  ``tests/test_oracle_pinning.py`` re-runs exactly those checks on
this restatement.  Nothing in the reference pins results tighter than 1e-6, for nu > 1e5
(quirk Q1) or for fp32: those are "parity unpinned" by the reference and pinned only here.

Decisions on reference quirks:
  Q1  studentt nu > 1e5 reproduces the reference's mis-imported Normal functions bug-for-bug.
  Q2  gamma.logpdf uses xlogy0(alpha-1, x) so that alpha == 1 gives 0 at x == 0 and -inf at
      x == +inf, which is what tests/test_gamma.py:14 pins (PyTensor folds the constant).
"""

from __future__ import annotations

import numpy as np
from scipy import linalg as sla
from scipy import special as sp

_F = np.float64
_LOG_SQRT_2PI = np.log(np.sqrt(2.0 * np.pi))


def _f(*arrays):
    return tuple(np.asarray(a, dtype=_F) for a in arrays)


def _quiet():
    return np.errstate(all="ignore")


# --------------------------------------------------------------------------- primitives
def xlogy0(a, y):
    """pytensor.tensor.xlogx.xlogy0: 0 where a == 0, else a*log(y)."""
    a, y = _f(a, y)
    with _quiet():
        return np.where(a == 0.0, 0.0, a * np.log(y))


def betaln(a, b):
    """pytensor.tensor.special.betaln: plain composition of three gammaln."""
    a, b = _f(a, b)
    return sp.gammaln(a) + sp.gammaln(b) - sp.gammaln(a + b)


def log1mexp(x):
    """pt.log1mexp: log(1 - exp(x)) for x <= 0."""
    (x,) = _f(x)
    with _quiet():
        return np.where(x < -np.log(2.0), np.log1p(-np.exp(x)), np.log(-np.expm1(x)))


def ppf_bounds_cont(x_val, q, lower, upper):
    """helper.py:29-54."""
    q = np.asarray(q, dtype=_F)
    inner = np.where(q == 0, lower, np.where(q == 1, upper, x_val))
    return np.where((q < 0) | (q > 1), np.nan, inner)


def ppf_bounds_disc(x_val, q, lower, upper):
    """helper.py:57-87."""
    q = np.asarray(q, dtype=_F)
    inner = np.where(q == 0, lower - 1, np.where(q == 1, upper, x_val))
    return np.where(q < 0, np.nan, np.where(q > 1, np.nan, inner))


def cdf_bounds(prob, x, lower, upper):
    """helper.py:4-26."""
    x = np.asarray(x, dtype=_F)
    return np.where(x < lower, 0.0, np.where(x >= upper, 1.0, prob))


def sf_bounds(x_val, q, lower, upper):
    """helper.py:90-91."""
    q = np.asarray(q, dtype=_F)
    return np.where(q < lower, 1.0, np.where(q > upper, 0.0, x_val))


def logdiffexp(a, b):
    """helper.py:94-96."""
    a, b = _f(a, b)
    return a + log1mexp(b - a)


# --------------------------------------------------------------------------- normal
def normal_logpdf(x, mu, sigma):
    """normal.py:79-80."""
    x, mu, sigma = _f(x, mu, sigma)
    with _quiet():
        return -0.5 * ((x - mu) / sigma) ** 2 - _LOG_SQRT_2PI - np.log(sigma)


def normal_logcdf(x, mu, sigma):
    """normal.py:70-76."""
    x, mu, sigma = _f(x, mu, sigma)
    with _quiet():
        z = (x - mu) / sigma
        left = np.log(sp.erfcx(-z / np.sqrt(2.0)) / 2.0) - z * z / 2.0
        right = np.log1p(-sp.erfc(z / np.sqrt(2.0)) / 2.0)
        return np.where(z < -1.0, left, right)


def normal_logsf(x, mu, sigma):
    """normal.py:83-84."""
    x, mu, sigma = _f(x, mu, sigma)
    return normal_logcdf(-x, -mu, sigma)


def normal_cdf(x, mu, sigma):
    """normal.py:46-47."""
    x, mu, sigma = _f(x, mu, sigma)
    with _quiet():
        return 0.5 * (1.0 + sp.erf((x - mu) / (sigma * 2**0.5)))


def normal_pdf(x, mu, sigma):
    """normal.py:54-55."""
    x, mu, sigma = _f(x, mu, sigma)
    with _quiet():
        return 1.0 / np.sqrt(2.0 * np.pi * sigma**2) * np.exp(-0.5 * ((x - mu) / sigma) ** 2)


def normal_sf(x, mu, sigma):
    """normal.py:62-63."""
    with _quiet():
        return np.exp(normal_logsf(x, mu, sigma))


def normal_ppf(q, mu, sigma):
    """normal.py:58-59."""
    q, mu, sigma = _f(q, mu, sigma)
    with _quiet():
        raw = mu + sigma * 2**0.5 * sp.erfinv(2.0 * q - 1.0)
    return ppf_bounds_cont(raw, q, -np.inf, np.inf)


def normal_isf(x, mu, sigma):
    """normal.py:50-51."""
    (x,) = _f(x)
    return normal_ppf(1.0 - x, mu, sigma)


# --------------------------------------------------------------------------- gamma
def gamma_logpdf(x, alpha, beta):
    """gamma.py:77-82 with quirk Q2 (xlogy0 for the (alpha-1)*log(x) term)."""
    x, alpha, beta = _f(x, alpha, beta)
    with _quiet():
        body = xlogy0(alpha - 1.0, x) - x * beta - sp.gammaln(alpha) + alpha * np.log(beta)
        return np.where(x < 0, -np.inf, body)


def gamma_logcdf(x, alpha, beta):
    """gamma.py:65-74."""
    x, alpha, beta = _f(x, alpha, beta)
    with _quiet():
        xb = x * beta
        lo = np.log(sp.gammainc(alpha, xb))
        hi = np.log1p(-sp.gammaincc(alpha, xb))
        return np.where(x < 0.0, -np.inf, np.where(xb < alpha, lo, hi))


def gamma_logsf(x, alpha, beta):
    """gamma.py:85-94."""
    x, alpha, beta = _f(x, alpha, beta)
    with _quiet():
        xb = x * beta
        lo = np.log1p(-sp.gammainc(alpha, xb))
        hi = np.log(sp.gammaincc(alpha, xb))
        return np.where(x < 0.0, 0.0, np.where(xb < alpha, lo, hi))


def gamma_pdf(x, alpha, beta):
    """gamma.py:44-45."""
    with _quiet():
        return np.exp(gamma_logpdf(x, alpha, beta))


def gamma_cdf(x, alpha, beta):
    """gamma.py:40-41."""
    with _quiet():
        return np.exp(gamma_logcdf(x, alpha, beta))


def gamma_sf(x, alpha, beta):
    """gamma.py:53-54."""
    return 1.0 - gamma_cdf(x, alpha, beta)


def gamma_ppf(q, alpha, beta):
    """gamma.py:48-50."""
    q, alpha, beta = _f(q, alpha, beta)
    with _quiet():
        raw = sp.gammaincinv(alpha, q) / beta
    return ppf_bounds_cont(raw, q, 0.0, np.inf)


def gamma_isf(x, alpha, beta):
    """gamma.py:57-58."""
    (x,) = _f(x)
    return gamma_ppf(1.0 - x, alpha, beta)


# --------------------------------------------------------------------------- beta
def beta_logpdf(x, alpha, beta):
    """beta.py:102-108."""
    x, alpha, beta = _f(x, alpha, beta)
    with _quiet():
        body = (xlogy0(alpha - 1.0, x) + xlogy0(beta - 1.0, 1.0 - x)) - (
            xlogy0(alpha + beta - 1.0, 1.0) + betaln(alpha, beta)
        )
        return np.where((x < 0) | (x > 1), -np.inf, body)


def beta_logcdf(x, alpha, beta):
    """beta.py:90-99."""
    x, alpha, beta = _f(x, alpha, beta)
    with _quiet():
        mid = np.log(sp.betainc(alpha, beta, x))
        return np.where(x < 0, -np.inf, np.where(x < 1, mid, 0.0))


def beta_logsf(x, alpha, beta):
    """beta.py:111-120."""
    x, alpha, beta = _f(x, alpha, beta)
    with _quiet():
        mid = np.log(sp.betainc(beta, alpha, 1.0 - x))
        return np.where(x < 0, 0.0, np.where(x < 1, mid, -np.inf))


def beta_pdf(x, alpha, beta):
    """beta.py:74-75."""
    with _quiet():
        return np.exp(beta_logpdf(x, alpha, beta))


def beta_cdf(x, alpha, beta):
    """beta.py:66-67."""
    with _quiet():
        return np.exp(beta_logcdf(x, alpha, beta))


def beta_sf(x, alpha, beta):
    """beta.py:82-83."""
    with _quiet():
        return np.exp(beta_logsf(x, alpha, beta))


def beta_ppf(q, alpha, beta):
    """beta.py:78-79."""
    q, alpha, beta = _f(q, alpha, beta)
    with _quiet():
        raw = sp.betaincinv(alpha, beta, q)
    return ppf_bounds_cont(raw, q, 0.0, 1.0)


def beta_isf(x, alpha, beta):
    """beta.py:70-71."""
    (x,) = _f(x)
    return beta_ppf(1.0 - x, alpha, beta)


# --------------------------------------------------------------------------- studentt
def studentt_logpdf(x, nu, mu, sigma):
    """studentt.py:108-119; the nu > 1e5 arm is normal.logcdf (mis-import, studentt.py:8)."""
    x, nu, mu, sigma = _f(x, nu, mu, sigma)
    with _quiet():
        exact = (
            sp.gammaln((nu + 1.0) / 2.0)
            - sp.gammaln(nu / 2.0)
            - 0.5 * np.log(nu * np.pi * sigma**2)
            - 0.5 * (nu + 1.0) * np.log1p((x - mu) ** 2 / (sigma**2 * nu))
        )
        return np.where(nu > 1e5, normal_logcdf(x, mu, sigma), exact)


def studentt_logcdf(x, nu, mu, sigma):
    """studentt.py:100-105; the nu > 1e5 arm is normal.cdf(z, mu, sigma) (mis-import, :6)."""
    x, nu, mu, sigma = _f(x, nu, mu, sigma)
    with _quiet():
        z = (x - mu) / sigma
        factor = 0.5 * sp.betainc(0.5 * nu, 0.5, nu / (z**2 + nu))
        exact = np.where(z < 0, np.log(factor), np.log1p(-factor))
        return np.where(nu > 1e5, normal_cdf(z, mu, sigma), exact)


def studentt_logsf(x, nu, mu, sigma):
    """studentt.py:122-123."""
    x, nu, mu, sigma = _f(x, nu, mu, sigma)
    return studentt_logcdf(-x, nu, -mu, sigma)


def studentt_pdf(x, nu, mu, sigma):
    """studentt.py:75-76."""
    with _quiet():
        return np.exp(studentt_logpdf(x, nu, mu, sigma))


def studentt_cdf(x, nu, mu, sigma):
    """studentt.py:67-68."""
    with _quiet():
        return np.exp(studentt_logcdf(x, nu, mu, sigma))


def studentt_sf(x, nu, mu, sigma):
    """studentt.py:92-93."""
    x, nu, mu, sigma = _f(x, nu, mu, sigma)
    return studentt_cdf(-x, nu, -mu, sigma)


def studentt_ppf(q, nu, mu, sigma):
    """studentt.py:79-89."""
    q, nu, mu, sigma = _f(q, nu, mu, sigma)
    with _quiet():
        lo = -np.sqrt(nu) * np.sqrt(1.0 / sp.betaincinv(nu * 0.5, 0.5, 2.0 * q) - 1.0)
        hi = np.sqrt(nu) * np.sqrt(1.0 / sp.betaincinv(nu * 0.5, 0.5, 2.0 * (1.0 - q)) - 1.0)
        r = np.where(np.abs(q - 0.5) < 1e-14, 0.0, np.where(q < 0.5, lo, hi))
        raw = mu + r * sigma
    return ppf_bounds_cont(raw, q, -np.inf, np.inf)


def studentt_isf(x, nu, mu, sigma):
    """studentt.py:71-72."""
    (x,) = _f(x)
    return studentt_ppf(1.0 - x, nu, mu, sigma)


# --------------------------------------------------------------------------- poisson
def poisson_logpdf(x, mu):
    """poisson.py:70-71 (no support switch, quirk Q6)."""
    x, mu = _f(x, mu)
    with _quiet():
        return xlogy0(x, mu) - sp.gammaln(x + 1.0) - mu


def poisson_cdf(x, mu):
    """poisson.py:49-50."""
    x, mu = _f(x, mu)
    with _quiet():
        return cdf_bounds(sp.gammaincc(x + 1.0, mu), x, 0, np.inf)


def poisson_sf(x, mu):
    """poisson.py:58-59."""
    x, mu = _f(x, mu)
    with _quiet():
        return sf_bounds(sp.gammainc(np.floor(x) + 1.0, mu), x, 0, np.inf)


def poisson_logcdf(x, mu):
    """poisson.py:74-75."""
    with _quiet():
        return np.log(poisson_cdf(x, mu))


def poisson_logsf(x, mu):
    """poisson.py:78-79."""
    with _quiet():
        return np.log(poisson_sf(x, mu))


def poisson_pdf(x, mu):
    """poisson.py:45-46."""
    with _quiet():
        return np.exp(poisson_logpdf(x, mu))


# --------------------------------------------------------------------------- binomial
def binomial_logpdf(x, n, p):
    """binomial.py:66-75."""
    x, n, p = _f(x, n, p)
    with _quiet():
        body = (
            sp.gammaln(n + 1.0)
            - sp.gammaln(x + 1.0)
            - sp.gammaln(n - x + 1.0)
            + xlogy0(x, p)
            + (n - x) * np.log1p(-p)
        )
        return np.where((x < 0) | (x > n), -np.inf, body)


def binomial_cdf(x, n, p):
    """binomial.py:45-46."""
    x, n, p = _f(x, n, p)
    with _quiet():
        return cdf_bounds(sp.betainc(n - x, x + 1.0, 1.0 - p), x, 0, n)


def binomial_sf(x, n, p):
    """binomial.py:54-55."""
    return 1.0 - binomial_cdf(x, n, p)


def binomial_logcdf(x, n, p):
    """binomial.py:78-79."""
    with _quiet():
        return np.log(binomial_cdf(x, n, p))


def binomial_logsf(x, n, p):
    """binomial.py:82-83."""
    with _quiet():
        return np.log1p(-binomial_cdf(x, n, p))


def binomial_pdf(x, n, p):
    """binomial.py:41-42."""
    with _quiet():
        return np.exp(binomial_logpdf(x, n, p))


# --------------------------------------------------------------------------- discrete ppf
def find_ppf_discrete(q, x0, lower, upper, cdf_func, pmf_func, *params, max_iter=100, tol=1e-7):
    """optimization.py:102-129: batch iteration with `until(all(found))`."""
    q = np.asarray(q, dtype=_F)
    x = np.floor(x0) + np.zeros_like(q)
    with _quiet():
        for _ in range(max_iter):
            x_int = np.floor(x)
            cdf_val = cdf_func(x_int, *params)
            cdf_minus = cdf_func(x_int - 1, *params)
            found = (cdf_val >= q - tol) & (cdf_minus < q)
            pmf_val = np.maximum(pmf_func(x_int, *params), 1e-10)
            delta = (cdf_val - q) / pmf_val
            delta_d = np.where(np.abs(delta) < 0.5, np.sign(cdf_val - q), np.floor(delta))
            x_new = x_int - delta_d
            x_new = np.where(x_new < lower, lower, np.where(x_new > upper, upper, x_new))  # pt.clip keeps NaN
            x = np.where(found, x_int, x_new)
            if np.all(found):
                break
    return ppf_bounds_disc(x, q, lower, upper)


def poisson_ppf(q, mu):
    """poisson.py:53-55."""
    return find_ppf_discrete(q, mu, 0, np.inf, poisson_cdf, poisson_pdf, mu)


def poisson_isf(q, mu):
    """poisson.py:62-63."""
    return poisson_ppf(1.0 - np.asarray(q, dtype=_F), mu)


def binomial_ppf(q, n, p):
    """binomial.py:49-51 (x0 = mean = n p)."""
    return find_ppf_discrete(q, n * p, 0, n, binomial_cdf, binomial_pdf, n, p)


def binomial_isf(q, n, p):
    """binomial.py:58-59."""
    return binomial_ppf(1.0 - np.asarray(q, dtype=_F), n, p)


# --------------------------------------------------------------------------- multivariate
def _chol_lower(cov):
    cov = np.asarray(cov, dtype=_F)
    if cov.ndim == 2:
        return sla.cholesky(cov, lower=True)
    flat = cov.reshape((-1,) + cov.shape[-2:])
    out = np.stack([sla.cholesky(c, lower=True) for c in flat])
    return out.reshape(cov.shape)


def logdet_from_cholesky(chol):
    """mvnormal.py:9-14."""
    diag = np.diagonal(chol, axis1=-2, axis2=-1)
    with _quiet():
        return 2.0 * np.sum(np.log(diag), axis=-1), np.all(diag > 0, axis=-1)


def quaddist_chol(value, mu, cov):
    """mvnormal.py:17-37 (one forward substitution per observation row)."""
    value = np.asarray(value, dtype=_F)
    if value.ndim == 0:
        raise ValueError("Value can't be a scalar")
    onedim = value.ndim == 1
    if onedim:
        value = value[None, :]
    chol = _chol_lower(cov)
    logdet, posdef = logdet_from_cholesky(chol)
    delta = value - np.asarray(mu, dtype=_F)
    if chol.ndim == 2:
        flat = delta.reshape(-1, delta.shape[-1])
        y = sla.solve_triangular(chol, flat.T, lower=True).T.reshape(delta.shape)
    else:
        # Blockwise over the leading batch axes of cov (b_ndim=1).
        bshape = np.broadcast_shapes(delta.shape[:-1], chol.shape[:-2])
        d_b = np.broadcast_to(delta, bshape + delta.shape[-1:]).reshape(-1, delta.shape[-1])
        c_b = np.broadcast_to(chol, bshape + chol.shape[-2:]).reshape((-1,) + chol.shape[-2:])
        y = np.stack(
            [sla.solve_triangular(c, d, lower=True) for c, d in zip(c_b, d_b)]
        ).reshape(bshape + delta.shape[-1:])
    quad = np.sum(y**2, axis=-1)
    if onedim:
        return quad[0], logdet, posdef
    return quad, logdet, posdef


def mvnormal_logpdf(x, mu, cov):
    """mvnormal.py:80-83 (posdef is ignored, quirk Q5)."""
    x = np.asarray(x, dtype=_F)
    quad, logdet, _ = quaddist_chol(x, mu, cov)
    k = float(x.shape[-1])
    return -0.5 * (k * np.log(2.0 * np.pi) + logdet + quad)


def mvstudentt_logpdf(x, nu, mu, cov):
    """mvstudentt.py:58-68 (the outer switch on x < -inf / x > inf is always false)."""
    x = np.asarray(x, dtype=_F)
    nu = np.asarray(nu, dtype=_F)
    quad, logdet, _ = quaddist_chol(x, mu, cov)
    k = float(x.shape[-1])
    norm = sp.gammaln((nu + k) / 2.0) - sp.gammaln(nu / 2.0) - 0.5 * k * np.log(nu * np.pi)
    inner = -(nu + k) / 2.0 * np.log1p(quad / nu)
    return norm + inner - 0.5 * logdet


def matrixnormal_logpdf(X, mu, rowcov, colcov):
    """matrixnormal.py:68-94."""
    X, mu, rowcov, colcov = _f(X, mu, rowcov, colcov)
    m, n = mu.shape[-2], mu.shape[-1]
    chol_r = _chol_lower(rowcov)
    chol_c = _chol_lower(colcov)
    logdet_r, _ = logdet_from_cholesky(chol_r)
    logdet_c, _ = logdet_from_cholesky(chol_c)
    delta = X - mu
    bshape = np.broadcast_shapes(delta.shape[:-2], chol_r.shape[:-2], chol_c.shape[:-2])
    d_b = np.broadcast_to(delta, bshape + (m, n)).reshape(-1, m, n)
    r_b = np.broadcast_to(chol_r, bshape + (m, m)).reshape(-1, m, m)
    c_b = np.broadcast_to(chol_c, bshape + (n, n)).reshape(-1, n, n)
    quad = np.empty(d_b.shape[0], dtype=_F)
    for i in range(d_b.shape[0]):
        y = sla.solve_triangular(r_b[i], d_b[i], lower=True)
        z = sla.solve_triangular(c_b[i], y.T, lower=True)
        quad[i] = np.sum(z**2)
    quad = quad.reshape(bshape)
    log_norm = 0.5 * m * n * np.log(2.0 * np.pi) + 0.5 * n * logdet_r + 0.5 * m * logdet_c
    return -0.5 * quad - log_norm


# --------------------------------------------------------------------------- categorical
def _normalize_p(p):
    p = np.asarray(p, dtype=_F)
    return p / np.sum(p, axis=-1, keepdims=True)


def categorical_logpdf(x, p):
    """categorical.py:64-74 (p is one probability vector, x any shape)."""
    p = _normalize_p(p)
    k = p.shape[-1]
    x = np.asarray(x, dtype=_F)
    with _quiet():
        x_int = x.astype(np.int64)  # C-style cast (truncation), as pt.cast does
        valid = (x >= 0) & (x < k) & (x == np.floor(x))
        safe = np.clip(x_int, 0, k - 1)
        return np.where(valid, np.log(p)[safe], -np.inf)


def categorical_cdf(x, p):
    """categorical.py:77-85."""
    p = _normalize_p(p)
    k = p.shape[-1]
    x = np.asarray(x, dtype=_F)
    with _quiet():
        safe = np.clip(np.floor(x), 0, k - 1).astype(np.int64)
        return cdf_bounds(np.cumsum(p, axis=-1)[safe], x, 0, k - 1)


def categorical_ppf(q, p):
    """categorical.py:100-108."""
    p = _normalize_p(p)
    k = p.shape[-1]
    q = np.asarray(q, dtype=_F)
    cs = np.cumsum(p, axis=-1)
    with _quiet():
        masked = np.where(cs >= q[..., None], np.arange(k), k)
        return ppf_bounds_disc(np.min(masked, axis=-1), q, 0, k - 1)


def categorical_pdf(x, p):
    """categorical.py:60-61."""
    return np.exp(categorical_logpdf(x, p))


def categorical_logcdf(x, p):
    """categorical.py:88-89."""
    with _quiet():
        return np.log(categorical_cdf(x, p))


def categorical_sf(x, p):
    """categorical.py:92-93."""
    return 1.0 - categorical_cdf(x, p)


def categorical_logsf(x, p):
    """categorical.py:96-97."""
    with _quiet():
        return np.log1p(-categorical_cdf(x, p))


def categorical_isf(q, p):
    """categorical.py:111-112."""
    return categorical_ppf(1.0 - np.asarray(q, dtype=_F), p)


# --------------------------------------------------------------------------- multinomial
def multinomial_logpdf(x, n, p):
    """multinomial.py:39-54."""
    x, n, p = _f(x, n, p)
    with _quiet():
        res = sp.gammaln(n + 1.0) - np.sum(sp.gammaln(x + 1.0), axis=-1)
        res = res + np.sum(x * np.log(np.maximum(p, 1e-10)), axis=-1)
        bad = np.any(x < 0, axis=-1) | (np.sum(x, axis=-1) != n)
        return np.where(bad, -np.inf, res)


# --------------------------------------------------------------------------- rvs (samplers)
def rvs(dist, params, size, seed):
    """Reference sampler semantics: pt.random.X -> numpy.random.Generator.X (normal.py:66-67,
    gamma.py:61-62 with scale=1/beta, beta.py:86-87, studentt.py:96-97, poisson.py:66-67,
    binomial.py:62-63).  Used as the second sample in two-sample KS / chi-square tests."""
    g = np.random.default_rng(seed)
    if dist == "normal":
        return g.normal(params[0], params[1], size)
    if dist == "gamma":
        return g.gamma(params[0], 1.0 / params[1], size)
    if dist == "beta":
        return g.beta(params[0], params[1], size)
    if dist == "studentt":
        return params[1] + params[2] * g.standard_t(params[0], size)
    if dist == "poisson":
        return g.poisson(params[0], size)
    if dist == "binomial":
        return g.binomial(int(params[0]), params[1], size)
    raise KeyError(dist)


TABLE = {
    name: obj
    for name, obj in list(globals().items())
    if callable(obj) and "_" in name and not name.startswith("_")
}

# closed-form / re-parameterised families: same table, separate file
from . import restatement_families as _families  # noqa: E402

TABLE.update(_families.FAMILY_TABLE)

# discrete families
from . import restatement_discrete as _discrete  # noqa: E402

TABLE.update(_discrete.DISCRETE_TABLE)

# solver families
from . import restatement_solver as _solver  # noqa: E402

TABLE.update(_solver.SOLVER_TABLE)
