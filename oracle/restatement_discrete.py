"""TEST INFRASTRUCTURE (not shipped, never imported by distributions_amd): CPU restatement, in
NumPy / SciPy float64, of the discrete families of the reference: bernoulli,
geometric, discreteuniform, discreteweibull, negativebinomial, betabinomial, hypergeometric,
zi_poisson, zi_binomial, zi_negativebinomial.  Each function restates the reference function of the
same name line by line (paths relative to /root/reference/pytensor_distributions), with
pt.X -> numpy / scipy.special.X and pt.switch -> np.where.  Pinned against the reference's own
SciPy test grids by tests/test_oracle_pinning.py where the reference tests against SciPy
(bernoulli, geometric, discreteuniform, negativebinomial, betabinomial, hypergeometric); the
reference tests discreteweibull and the zero-inflated families only empirically (samples, sum of
pmf, cdf + sf = 1): for those the restatement is pinned by the same identities, "parity unpinned"
by any reference value."""

from __future__ import annotations

import numpy as np
from scipy import special as sp

from .restatement import (_F, _f, _quiet, binomial_cdf, binomial_logpdf, cdf_bounds, find_ppf_discrete, poisson_cdf,
                          poisson_logpdf, poisson_sf, ppf_bounds_disc, sf_bounds, xlogy0)

inf = np.inf
sw = np.where


def _wrap(fn):
    def g(*args):
        with _quiet():
            return np.asarray(fn(*_f(*args)), dtype=_F)

    g.__name__ = fn.__name__
    return g


# --------------------------------------------------------------------------------- bernoulli.py
@_wrap
def bernoulli_cdf(x, p):
    return cdf_bounds(sw(x < 1, 1 - p, 1.0), x, 0, 1)


@_wrap
def bernoulli_ppf(q, p):
    return ppf_bounds_disc(sw(q <= 1 - p, 0, 1), q, 0, 1)


@_wrap
def bernoulli_sf(x, p):
    return sw(x < 0, 1.0, sw(x < 1, p, 0.0))


@_wrap
def bernoulli_isf(q, p):
    return bernoulli_ppf(1 - q, p)


@_wrap
def bernoulli_pdf(x, p):
    return sw(x == 1, p, sw(x == 0, 1 - p, 0.0))


@_wrap
def bernoulli_logpdf(x, p):
    return sw(x == 1, np.log(p), sw(x == 0, np.log(1 - p), -inf))


@_wrap
def bernoulli_logcdf(x, p):
    return sw(x < 0, -inf, sw(x < 1, np.log(1 - p), 0.0))


@_wrap
def bernoulli_logsf(x, p):
    return np.log1p(-bernoulli_cdf(x, p))


# --------------------------------------------------------------------------------- geometric.py
@_wrap
def geometric_cdf(x, p):
    k = np.floor(x)
    return cdf_bounds(-np.expm1(np.log1p(-p) * k), x, 1, inf)


@_wrap
def geometric_logpdf(x, p):
    return sw(x < 1, -inf, (x - 1) * np.log1p(-p) + np.log(p))


@_wrap
def geometric_pdf(x, p):
    return np.exp(geometric_logpdf(x, p))


@_wrap
def geometric_ppf(q, p):
    vals = np.ceil(np.log1p(-q) / np.log1p(-p))
    temp = geometric_cdf(vals - 1, p)
    return ppf_bounds_disc(sw((temp >= q) & (vals > 0), vals - 1, vals), q, 1, inf)


@_wrap
def geometric_sf(x, p):
    return 1.0 - geometric_cdf(x, p)


@_wrap
def geometric_isf(q, p):
    return geometric_ppf(1.0 - q, p)


@_wrap
def geometric_logcdf(x, p):
    return np.log(geometric_cdf(x, p))


@_wrap
def geometric_logsf(x, p):
    return np.log1p(-geometric_cdf(x, p))


# --------------------------------------------------------------------------- discreteuniform.py
@_wrap
def discreteuniform_logpdf(x, lower, upper):
    n = upper - lower + 1
    return sw((x >= lower) & (x <= upper), -np.log(n), -inf)


@_wrap
def discreteuniform_pdf(x, lower, upper):
    return np.exp(discreteuniform_logpdf(x, lower, upper))


@_wrap
def discreteuniform_cdf(x, lower, upper):
    return cdf_bounds((np.floor(x) - lower + 1) / (upper - lower + 1), x, lower, upper)


@_wrap
def discreteuniform_ppf(q, lower, upper):
    n = upper - lower + 1
    return ppf_bounds_disc(lower + np.ceil(q * n) - 1, q, lower, upper)


@_wrap
def discreteuniform_sf(x, lower, upper):
    return 1.0 - discreteuniform_cdf(x, lower, upper)


@_wrap
def discreteuniform_isf(q, lower, upper):
    return discreteuniform_ppf(1.0 - q, lower, upper)


@_wrap
def discreteuniform_logcdf(x, lower, upper):
    return np.log(discreteuniform_cdf(x, lower, upper))


@_wrap
def discreteuniform_logsf(x, lower, upper):
    return np.log1p(-discreteuniform_cdf(x, lower, upper))


# --------------------------------------------------------------------------- discreteweibull.py
@_wrap
def discreteweibull_cdf(x, q, beta):
    return cdf_bounds(1 - q ** ((x + 1) ** beta), x, 0, inf)


@_wrap
def discreteweibull_pdf(x, q, beta):
    return sw(x < 0, 0, q ** (np.floor(x) ** beta) - q ** ((np.floor(x) + 1) ** beta))


@_wrap
def discreteweibull_ppf(p, q, beta):
    return ppf_bounds_disc(np.ceil((np.log(1 - p) / np.log(q)) ** (1 / beta) - 1), p, 0, inf)


@_wrap
def discreteweibull_isf(x, q, beta):
    return discreteweibull_ppf(1 - x, q, beta)


@_wrap
def discreteweibull_sf(x, q, beta):
    return sw(x < 0, 1, q ** ((np.floor(x) + 1) ** beta))


@_wrap
def discreteweibull_logcdf(x, q, beta):
    return sw(x < 0, -inf, np.log(1 - q ** ((np.floor(x) + 1) ** beta)))


@_wrap
def discreteweibull_logpdf(x, q, beta):
    return sw(x < 0, -inf, np.log(q ** (np.floor(x) ** beta) - q ** ((np.floor(x) + 1) ** beta)))


@_wrap
def discreteweibull_logsf(x, q, beta):
    return sw(x < 0, 0, np.log(q ** ((np.floor(x) + 1) ** beta)))


# -------------------------------------------------------------------------- negativebinomial.py
@_wrap
def negativebinomial_logpdf(x, n, p):
    return sw(x < 0, -inf, sp.gammaln(x + n) - sp.gammaln(n) - sp.gammaln(x + 1) + xlogy0(n, p) + xlogy0(x, 1 - p))


@_wrap
def negativebinomial_pdf(x, n, p):
    return np.exp(negativebinomial_logpdf(x, n, p))


@_wrap
def negativebinomial_cdf(x, n, p):
    k = np.floor(x)
    return cdf_bounds(sp.betainc(n, k + 1, p), k, 0, inf)


@_wrap
def negativebinomial_sf(x, n, p):
    k = np.floor(x)
    return sf_bounds(sw(np.isinf(k), 0.0, sp.betainc(k + 1, n, 1 - p)), k, 0, inf)


@_wrap
def negativebinomial_ppf(q, n, p):
    return find_ppf_discrete(q, n * (1 - p) / p, 0, inf, negativebinomial_cdf, negativebinomial_pdf, n, p)


@_wrap
def negativebinomial_isf(q, n, p):
    return negativebinomial_ppf(1.0 - q, n, p)


@_wrap
def negativebinomial_logcdf(x, n, p):
    return np.log(negativebinomial_cdf(x, n, p))


@_wrap
def negativebinomial_logsf(x, n, p):
    return np.log(negativebinomial_sf(x, n, p))


# ------------------------------------------------------------------------------ betabinomial.py
@_wrap
def betabinomial_logpdf(x, n, alpha, beta):
    return sw((x < 0) | (x > n), -inf,
              sp.gammaln(n + 1) - sp.gammaln(x + 1) - sp.gammaln(n - x + 1) + sp.gammaln(x + alpha)
              + sp.gammaln(n - x + beta) - sp.gammaln(n + alpha + beta) + sp.gammaln(alpha + beta)
              - sp.gammaln(alpha) - sp.gammaln(beta))


@_wrap
def betabinomial_pdf(x, n, alpha, beta):
    return np.exp(betabinomial_logpdf(x, n, alpha, beta))


@_wrap
def betabinomial_cdf(x, n, alpha, beta):
    """betabinomial.py:87-95: sum over k = 0 .. max(x) of pdf(k) where k <= floor(x).  (The reference
    builds arange(0, max(x) + 1): infinite / NaN x make that fail there; here they are capped at n,
    beyond which every term is zero.)"""
    x, n, alpha, beta = np.broadcast_arrays(x, n, alpha, beta)
    finite = np.where(np.isfinite(x), x, 0.0)
    top = int(max(0.0, min(float(np.max(finite, initial=0.0)), float(np.max(n)))))
    k = np.arange(0, top + 1, dtype=_F).reshape((-1,) + (1,) * x.ndim)
    prob = np.sum(np.where(k <= np.floor(x), betabinomial_pdf(k, n, alpha, beta), 0.0), axis=0)
    return cdf_bounds(prob, x, 0, n)


@_wrap
def betabinomial_ppf(q, n, alpha, beta):
    return find_ppf_discrete(q, n * alpha / (alpha + beta), 0, n, betabinomial_cdf, betabinomial_pdf, n, alpha, beta)


@_wrap
def betabinomial_sf(x, n, alpha, beta):
    return 1 - betabinomial_cdf(x, n, alpha, beta)


@_wrap
def betabinomial_isf(x, n, alpha, beta):
    return betabinomial_ppf(1 - x, n, alpha, beta)


@_wrap
def betabinomial_logcdf(x, n, alpha, beta):
    return np.log(betabinomial_cdf(x, n, alpha, beta))


@_wrap
def betabinomial_logsf(x, n, alpha, beta):
    return np.log(betabinomial_sf(x, n, alpha, beta))


# ---------------------------------------------------------------------------- hypergeometric.py
def _log_binomial(n, k):
    return sp.gammaln(n + 1) - sp.gammaln(k + 1) - sp.gammaln(n - k + 1)


@_wrap
def hypergeometric_logpdf(x, N, k, n):
    x = np.floor(x)
    lower, upper = np.maximum(0, n + k - N), np.minimum(k, n)
    res = _log_binomial(k, x) + _log_binomial(N - k, n - x) - _log_binomial(N, n)
    return sw((x >= lower) & (x <= upper), res, -inf)


@_wrap
def hypergeometric_pdf(x, N, k, n):
    return np.exp(hypergeometric_logpdf(x, N, k, n))


@_wrap
def hypergeometric_cdf(x, N, k, n):
    """hypergeometric.py:76-90 (scalar N, k, n as in the reference's tests)."""
    lower, upper = np.maximum(0, n + k - N), np.minimum(k, n)
    safe = np.clip(np.floor(x), 0, upper)
    safe = sw(np.isnan(x), 0, safe)
    xv = np.arange(0, int(np.max(upper)) + 1, dtype=_F)
    cumsum = np.cumsum(np.exp(hypergeometric_logpdf(xv, np.max(N), np.max(k), np.max(n))))
    raw = cumsum[safe.astype(np.int64)]
    raw = sw(np.isnan(x), np.nan, raw)
    return cdf_bounds(raw, x, lower, upper)


@_wrap
def hypergeometric_logcdf(x, N, k, n):
    return np.log(hypergeometric_cdf(x, N, k, n))


@_wrap
def hypergeometric_sf(x, N, k, n):
    return 1.0 - hypergeometric_cdf(x, N, k, n)


@_wrap
def hypergeometric_logsf(x, N, k, n):
    return np.log1p(-hypergeometric_cdf(x, N, k, n))


@_wrap
def hypergeometric_ppf(q, N, k, n):
    lower, upper = np.maximum(0, n + k - N), np.minimum(k, n)
    return find_ppf_discrete(q, n * k / N, lower, upper, hypergeometric_cdf, hypergeometric_pdf, N, k, n)


@_wrap
def hypergeometric_isf(q, N, k, n):
    return hypergeometric_ppf(1.0 - q, N, k, n)


# -------------------------------------------------------------------------------- zi_poisson.py
@_wrap
def zi_poisson_logpdf(x, psi, mu):
    log_zero = np.log((1 - psi) + psi * np.exp(-mu))
    res = sw((x < 0) | np.isinf(x), -inf, sw(x == 0, log_zero, np.log(psi) + poisson_logpdf(x, mu)))
    return sw(np.isinf(x) & (x > 0), np.nan, res)


@_wrap
def zi_poisson_pdf(x, psi, mu):
    return sw(np.isinf(x) & (x > 0), np.nan, np.exp(zi_poisson_logpdf(x, psi, mu)))


@_wrap
def zi_poisson_cdf(x, psi, mu):
    return cdf_bounds((1 - psi) + psi * poisson_cdf(x, mu), x, 0, inf)


@_wrap
def zi_poisson_ppf(q, psi, mu):
    return find_ppf_discrete(q, 1 * mu, 0, inf, zi_poisson_cdf, zi_poisson_pdf, psi, mu)


@_wrap
def zi_poisson_sf(x, psi, mu):
    return sf_bounds(psi * poisson_sf(x, mu), x, 0, inf)


@_wrap
def zi_poisson_isf(q, psi, mu):
    return zi_poisson_ppf(1.0 - q, psi, mu)


@_wrap
def zi_poisson_logcdf(x, psi, mu):
    res = np.log1p(psi * (poisson_cdf(x, mu) - 1))
    return sw((x < 0) | np.isinf(x), sw(x < 0, -inf, 0.0), res)


@_wrap
def zi_poisson_logsf(x, psi, mu):
    return np.log(zi_poisson_sf(x, psi, mu))


# ------------------------------------------------------------------------------- zi_binomial.py
@_wrap
def zi_binomial_logpdf(x, psi, n, p):
    log_zero = np.log((1 - psi) + psi * np.power(1 - p, n))
    return sw((x < 0) | (x > n), -inf, sw(x == 0, log_zero, np.log(psi) + binomial_logpdf(x, n, p)))


@_wrap
def zi_binomial_pdf(x, psi, n, p):
    return np.exp(zi_binomial_logpdf(x, psi, n, p))


@_wrap
def zi_binomial_cdf(x, psi, n, p):
    return cdf_bounds((1 - psi) + psi * binomial_cdf(x, n, p), x, 0, n)


@_wrap
def zi_binomial_ppf(q, psi, n, p):
    return find_ppf_discrete(q, psi * n * p, 0, n, zi_binomial_cdf, zi_binomial_pdf, psi, n, p)


@_wrap
def zi_binomial_sf(x, psi, n, p):
    return 1.0 - zi_binomial_cdf(x, psi, n, p)


@_wrap
def zi_binomial_isf(q, psi, n, p):
    return zi_binomial_ppf(1.0 - q, psi, n, p)


@_wrap
def zi_binomial_logcdf(x, psi, n, p):
    res = np.log1p(psi * (binomial_cdf(x, n, p) - 1))
    return sw(x < 0, -inf, sw(x > n, 0.0, res))


@_wrap
def zi_binomial_logsf(x, psi, n, p):
    return np.log1p(-zi_binomial_cdf(x, psi, n, p))


# ----------------------------------------------------------------------- zi_negativebinomial.py
@_wrap
def zi_negativebinomial_logpdf(x, psi, n, p):
    log_zero = np.log((1 - psi) + psi * np.power(p, n))
    res = sw((x < 0) | np.isinf(x), -inf, sw(x == 0, log_zero, np.log(psi) + negativebinomial_logpdf(x, n, p)))
    return sw(np.isinf(x) & (x > 0), np.nan, res)


@_wrap
def zi_negativebinomial_pdf(x, psi, n, p):
    return sw(np.isinf(x) & (x > 0), np.nan, np.exp(zi_negativebinomial_logpdf(x, psi, n, p)))


@_wrap
def zi_negativebinomial_cdf(x, psi, n, p):
    return cdf_bounds((1 - psi) + psi * negativebinomial_cdf(x, n, p), x, 0, inf)


@_wrap
def zi_negativebinomial_ppf(q, psi, n, p):
    return find_ppf_discrete(q, psi * n * (1 - p) / p, 0, inf, zi_negativebinomial_cdf, zi_negativebinomial_pdf, psi, n, p)


@_wrap
def zi_negativebinomial_sf(x, psi, n, p):
    return sf_bounds(psi * negativebinomial_sf(x, n, p), x, 0, inf)


@_wrap
def zi_negativebinomial_isf(q, psi, n, p):
    return zi_negativebinomial_ppf(1.0 - q, psi, n, p)


@_wrap
def zi_negativebinomial_logcdf(x, psi, n, p):
    res = np.log1p(psi * (negativebinomial_cdf(x, n, p) - 1))
    return sw((x < 0) | np.isinf(x), sw(x < 0, -inf, 0.0), res)


@_wrap
def zi_negativebinomial_logsf(x, psi, n, p):
    return np.log(zi_negativebinomial_sf(x, psi, n, p))


DISCRETE_TABLE = {
    name: obj
    for name, obj in list(globals().items())
    if callable(obj) and "_" in name and not name.startswith("_") and name.split("_")[-1] in
    ("pdf", "logpdf", "cdf", "logcdf", "sf", "logsf", "ppf", "isf")
    and name.rsplit("_", 1)[0] in ("bernoulli", "geometric", "discreteuniform", "discreteweibull", "negativebinomial",
                                   "betabinomial", "hypergeometric", "zi_poisson", "zi_binomial", "zi_negativebinomial")
}
