"""TEST INFRASTRUCTURE (not shipped, never imported by distributions_amd): CPU restatement, in
NumPy / SciPy float64, of the closed-form and re-parameterised continuous families of the reference
(SURVEY.md 8(f) rank 1).  Each function restates the reference function of the same name line by
line (file:line cited; paths relative to /root/reference/pytensor_distributions), with
pt.X -> numpy / scipy.special.X and pt.switch -> np.where (both arms evaluated, as in the
reference).  Pinned against the reference's own SciPy test grids by tests/test_oracle_pinning.py
(tests/test_<dist>.py of the reference compare to exactly these scipy.stats distributions);
logitnormal has no SciPy counterpart in the reference's tests (empirical harness only) and is
pinned here by identities: parity unpinned by the reference for that family."""

from __future__ import annotations

import numpy as np
from scipy import special as sp

from .restatement import (_F, _f, _quiet, betaln, cdf_bounds, log1mexp, normal_ppf, ppf_bounds_cont, sf_bounds,
                          xlogy0)

inf = np.inf
sw = np.where


def log1pexp(x):
    """pt.log1pexp (softplus), PyTensor's piecewise form."""
    x = np.asarray(x, dtype=_F)
    with _quiet():
        return sw(x < -37.0, np.exp(x), sw(x < 18.0, np.log1p(np.exp(x)), sw(x < 33.3, x + np.exp(-x), x)))


def sigmoid(x):
    with _quiet():
        return 1.0 / (1.0 + np.exp(-np.asarray(x, dtype=_F)))


def _wrap(fn):
    def g(*args):
        with _quiet():
            return np.asarray(fn(*_f(*args)), dtype=_F)

    g.__name__ = fn.__name__
    g.__doc__ = fn.__doc__
    return g


# ------------------------------------------------------------------------------- exponential.py
@_wrap
def exponential_cdf(x, lam):  # :41-43
    return cdf_bounds(1.0 - np.exp(-lam * x), x, 0.0, inf)


@_wrap
def exponential_pdf(x, lam):  # :50-51
    return sw(x < 0.0, 0.0, lam * np.exp(-lam * x))


@_wrap
def exponential_ppf(q, lam):  # :54-56
    return ppf_bounds_cont(-np.log(1.0 - q) / lam, q, 0.0, inf)


@_wrap
def exponential_isf(x, lam):  # :46-47
    return exponential_ppf(1 - x, lam)


@_wrap
def exponential_sf(x, lam):  # :59-60
    return sw(x < 0.0, 1.0, np.exp(-lam * x))


@_wrap
def exponential_logcdf(x, lam):  # :67-68
    return sw(x < 0.0, -inf, np.log1p(-np.exp(-lam * x)))


@_wrap
def exponential_logpdf(x, lam):  # :71-72
    return sw(x < 0.0, -inf, np.log(lam) - lam * x)


@_wrap
def exponential_logsf(x, lam):  # :75-76
    return sw(x < 0.0, 0.0, -lam * x)


# --------------------------------------------------------------------------------- lognormal.py
@_wrap
def lognormal_cdf(x, mu, sigma):  # :45-47
    return cdf_bounds(0.5 * (1 + sp.erf((np.log(x) - mu) / (sigma * np.sqrt(2)))), x, 0, inf)


@_wrap
def lognormal_logpdf(x, mu, sigma):  # :83-91
    return sw(x <= 0, -inf,
              -np.log(x) - np.log(sigma) - 0.5 * np.log(2 * np.pi) - 0.5 * ((np.log(x) - mu) / sigma) ** 2)


@_wrap
def lognormal_pdf(x, mu, sigma):  # :54-55
    return np.exp(lognormal_logpdf(x, mu, sigma))


@_wrap
def lognormal_ppf(q, mu, sigma):  # :58-59
    return ppf_bounds_cont(np.exp(mu + sigma * np.sqrt(2) * sp.erfinv(2 * q - 1)), q, 0, inf)


@_wrap
def lognormal_isf(x, mu, sigma):  # :50-51
    return lognormal_ppf(1 - x, mu, sigma)


@_wrap
def lognormal_logcdf(x, mu, sigma):  # :70-80
    z = (np.log(x) - mu) / sigma
    return sw(x <= 0, -inf, sw(z < -1.0, np.log(sp.erfcx(-z / np.sqrt(2.0)) / 2.0) - np.square(z) / 2.0,
                               np.log1p(-sp.erfc(z / np.sqrt(2.0)) / 2.0)))


@_wrap
def lognormal_logsf(x, mu, sigma):  # :94-104
    z = (np.log(x) - mu) / sigma
    return sw(x <= 0, 0.0, sw(z > 1.0, np.log(sp.erfcx(z / np.sqrt(2.0)) / 2.0) - np.square(z) / 2.0,
                              np.log1p(-0.5 * (1 + sp.erf(z / np.sqrt(2.0))))))


@_wrap
def lognormal_sf(x, mu, sigma):  # :62-63
    return np.exp(lognormal_logsf(x, mu, sigma))


# -------------------------------------------------------------------------------- halfnormal.py
@_wrap
def halfnormal_cdf(x, sigma):  # :38-39
    return sw(x < 0, 0.0, sp.erf(x / (sigma * np.sqrt(2))))


@_wrap
def halfnormal_logcdf(x, sigma):  # :42-43
    return sw(x < 0, -inf, np.log(sp.erf(x / (sigma * np.sqrt(2)))))


@_wrap
def halfnormal_logpdf(x, sigma):  # :67-70
    return sw(x < 0, -inf, np.log(np.sqrt(2 / np.pi)) - np.log(sigma) - 0.5 * (x / sigma) ** 2)


@_wrap
def halfnormal_pdf(x, sigma):  # :50-51
    return np.exp(halfnormal_logpdf(x, sigma))


@_wrap
def halfnormal_ppf(q, sigma):  # :54-56
    return ppf_bounds_cont(sigma * np.sqrt(2) * sp.erfinv(q), q, 0, inf)


@_wrap
def halfnormal_isf(x, sigma):  # :46-47
    return halfnormal_ppf(1 - x, sigma)


@_wrap
def halfnormal_sf(x, sigma):  # :59-60
    return 1 - halfnormal_cdf(x, sigma)


@_wrap
def halfnormal_logsf(x, sigma):  # :73-74
    return log1mexp(halfnormal_logcdf(x, sigma))


# ------------------------------------------------------------------------------------ cauchy.py
@_wrap
def cauchy_cdf(x, alpha, beta):  # :44-45
    return 1 / np.pi * np.arctan((x - alpha) / beta) + 0.5


@_wrap
def cauchy_logpdf(x, alpha, beta):  # :72-73
    return -np.log(np.pi) - np.log(beta) - np.log(1 + ((x - alpha) / beta) ** 2)


@_wrap
def cauchy_pdf(x, alpha, beta):  # :52-53
    return np.exp(cauchy_logpdf(x, alpha, beta))


@_wrap
def cauchy_ppf(q, alpha, beta):  # :56-57
    return ppf_bounds_cont(alpha + beta * np.tan(np.pi * (q - 0.5)), q, -inf, inf)


@_wrap
def cauchy_isf(x, alpha, beta):  # :48-49
    return cauchy_ppf(1 - x, alpha, beta)


@_wrap
def cauchy_logcdf(x, alpha, beta):  # :68-69
    return np.log(np.clip(cauchy_cdf(x, alpha, beta), 0, 1))


@_wrap
def cauchy_logsf(x, alpha, beta):  # :76-78
    return np.log(np.clip(0.5 - (1 / np.pi) * np.arctan((x - alpha) / beta), 0, 1))


@_wrap
def cauchy_sf(x, alpha, beta):  # :60-61
    return np.exp(cauchy_logsf(x, alpha, beta))


# -------------------------------------------------------------------------------- halfcauchy.py
@_wrap
def halfcauchy_logcdf(x, beta):  # :42-47
    return sw(x < 0, -inf, np.log(2 * np.arctan(x / beta) / np.pi))


@_wrap
def halfcauchy_cdf(x, beta):  # :38-39
    return np.exp(halfcauchy_logcdf(x, beta))


@_wrap
def halfcauchy_logpdf(x, beta):  # :72-75
    return sw(x < 0, -inf, np.log(2) - np.log(np.pi * beta) - np.log(1 + (x / beta) ** 2))


@_wrap
def halfcauchy_pdf(x, beta):  # :54-55
    return np.exp(halfcauchy_logpdf(x, beta))


@_wrap
def halfcauchy_ppf(q, beta):  # :58-60
    return ppf_bounds_cont(beta * np.tan(np.pi / 2 * q), q, 0, inf)


@_wrap
def halfcauchy_isf(x, beta):  # :50-51
    return halfcauchy_ppf(1 - x, beta)


@_wrap
def halfcauchy_sf(x, beta):  # :63-64
    return 1 - halfcauchy_cdf(x, beta)


@_wrap
def halfcauchy_logsf(x, beta):  # :78-79
    return log1mexp(halfcauchy_logcdf(x, beta))


# ----------------------------------------------------------------------------------- laplace.py
@_wrap
def laplace_pdf(x, mu, b):
    return 0.5 / b * np.exp(-np.abs(x - mu) / b)


@_wrap
def laplace_cdf(x, mu, b):
    r = sw(x < mu, 0.5 * np.exp((x - mu) / b), 1.0 - 0.5 * np.exp(-(x - mu) / b))
    return cdf_bounds(r, x, -inf, inf)


@_wrap
def laplace_sf(x, mu, b):
    return 1.0 - laplace_cdf(x, mu, b)


@_wrap
def laplace_ppf(q, mu, b):
    r = sw(q < 0.5, mu + b * np.log(2.0 * q), mu - b * np.log(2.0 * (1.0 - q)))
    return ppf_bounds_cont(r, q, -inf, inf)


@_wrap
def laplace_isf(q, mu, b):
    return laplace_ppf(1.0 - q, mu, b)


@_wrap
def laplace_logpdf(x, mu, b):
    return np.log(0.5) - np.abs((x - mu) / b) - np.log(b)


@_wrap
def laplace_logcdf(x, mu, b):
    y = (x - mu) / b
    return sw(y <= 0, np.log(0.5) + y, sw(y > 1, np.log1p(-0.5 * np.exp(-y)), np.log(1 - 0.5 * np.exp(-y))))


@_wrap
def laplace_logsf(x, mu, b):
    return laplace_logcdf(-x, -mu, b)


# ---------------------------------------------------------------------------------- logistic.py
@_wrap
def logistic_cdf(x, mu, s):
    return 1 / (1 + np.exp(-(x - mu) / s))


@_wrap
def logistic_logpdf(x, mu, s):
    z = (x - mu) / s
    return sw(x == -inf, -inf, -np.log(s) - z - 2.0 * np.log1p(np.exp(-z)))


@_wrap
def logistic_pdf(x, mu, s):
    return np.exp(logistic_logpdf(x, mu, s))


@_wrap
def logistic_ppf(q, mu, s):
    return ppf_bounds_cont(mu + s * np.log(q / (1 - q)), q, -inf, inf)


@_wrap
def logistic_isf(x, mu, s):
    return logistic_ppf(1 - x, mu, s)


@_wrap
def logistic_logcdf(x, mu, s):
    return -log1pexp(-(x - mu) / s)


@_wrap
def logistic_logsf(x, mu, s):
    return -log1pexp((x - mu) / s)


@_wrap
def logistic_sf(x, mu, s):
    return np.exp(logistic_logsf(x, mu, s))


# ------------------------------------------------------------------------------------ gumbel.py
@_wrap
def gumbel_cdf(x, mu, beta):
    return cdf_bounds(np.exp(-np.exp(-(x - mu) / beta)), x, -inf, inf)


@_wrap
def gumbel_logpdf(x, mu, beta):
    z = (x - mu) / beta
    return -z - np.exp(-z) - np.log(beta)


@_wrap
def gumbel_pdf(x, mu, beta):
    return np.exp(gumbel_logpdf(x, mu, beta))


@_wrap
def gumbel_ppf(q, mu, beta):
    return ppf_bounds_cont(mu - beta * np.log(-np.log(q)), q, -inf, inf)


@_wrap
def gumbel_isf(x, mu, beta):
    return gumbel_ppf(1 - x, mu, beta)


@_wrap
def gumbel_logcdf(x, mu, beta):
    return -np.exp(-(x - mu) / beta)


@_wrap
def gumbel_logsf(x, mu, beta):
    return np.log1p(-np.exp(-np.exp(-(x - mu) / beta)))


@_wrap
def gumbel_sf(x, mu, beta):
    return np.exp(gumbel_logsf(x, mu, beta))


# ----------------------------------------------------------------------------------- weibull.py
@_wrap
def weibull_logcdf(x, alpha, beta):
    a = (x / beta) ** alpha
    return sw(x < 0, -inf, log1mexp(-a))


@_wrap
def weibull_cdf(x, alpha, beta):
    return np.exp(weibull_logcdf(x, alpha, beta))


@_wrap
def weibull_logpdf(x, alpha, beta):
    return sw(x < 0, -inf, np.log(alpha / beta) + xlogy0(alpha - 1, x / beta) - (x / beta) ** alpha)


@_wrap
def weibull_pdf(x, alpha, beta):
    return np.exp(weibull_logpdf(x, alpha, beta))


@_wrap
def weibull_ppf(q, alpha, beta):
    return ppf_bounds_cont(beta * (-np.log(1 - q)) ** (1 / alpha), q, 0.0, inf)


@_wrap
def weibull_isf(x, alpha, beta):
    return weibull_ppf(1 - x, alpha, beta)


@_wrap
def weibull_logsf(x, alpha, beta):
    return sw(x < 0, 0, -((x / beta) ** alpha))


@_wrap
def weibull_sf(x, alpha, beta):
    return np.exp(weibull_logsf(x, alpha, beta))


# ------------------------------------------------------------------------------------ pareto.py
@_wrap
def pareto_cdf(x, alpha, m):
    return sw(x < m, np.zeros_like(x), 1 - (m / x) ** alpha)


@_wrap
def pareto_logpdf(x, alpha, m):
    return sw(x < m, -inf, np.log(alpha) + alpha * np.log(m) - (alpha + 1) * np.log(x))


@_wrap
def pareto_pdf(x, alpha, m):
    return np.exp(pareto_logpdf(x, alpha, m))


@_wrap
def pareto_ppf(q, alpha, m):
    return ppf_bounds_cont(m * (1 - q) ** (-1 / alpha), q, m, inf)


@_wrap
def pareto_sf(x, alpha, m):
    return 1 - pareto_cdf(x, alpha, m)


@_wrap
def pareto_isf(x, alpha, m):
    return pareto_ppf(1 - x, alpha, m)


@_wrap
def pareto_logcdf(x, alpha, m):
    return sw(x < m, -inf, np.log(1 - (m / x) ** alpha))


@_wrap
def pareto_logsf(x, alpha, m):
    return np.log(pareto_sf(x, alpha, m))


# ----------------------------------------------------------------------------------- uniform.py
@_wrap
def uniform_cdf(x, lower, upper):
    return sw(x < lower, np.zeros_like(x), sw(x > upper, np.ones_like(x), (x - lower) / (upper - lower)))


@_wrap
def uniform_logpdf(x, lower, upper):
    return sw((x < lower) | (x > upper), -inf, -np.log(upper - lower))


@_wrap
def uniform_pdf(x, lower, upper):
    return np.exp(uniform_logpdf(x, lower, upper))


@_wrap
def uniform_ppf(q, lower, upper):
    return ppf_bounds_cont(lower + q * (upper - lower), q, lower, upper)


@_wrap
def uniform_sf(x, lower, upper):
    return 1 - uniform_cdf(x, lower, upper)


@_wrap
def uniform_isf(x, lower, upper):
    return uniform_ppf(1 - x, lower, upper)


@_wrap
def uniform_logcdf(x, lower, upper):
    return sw(x < lower, -inf, sw(x > upper, 0, np.log((x - lower) / (upper - lower))))


@_wrap
def uniform_logsf(x, lower, upper):
    return np.log(uniform_sf(x, lower, upper))


# -------------------------------------------------------------------------------- triangular.py
@_wrap
def triangular_pdf(x, lower, c, upper):
    res = sw(x < c, 2 * (x - lower) / ((upper - lower) * (c - lower)), 2 * (upper - x) / ((upper - lower) * (upper - c)))
    return sw((lower <= x) & (x <= upper), res, 0.0)


@_wrap
def triangular_cdf(x, lower, c, upper):
    return sw(x <= lower, 0.0,
              sw(x <= c, ((x - lower) ** 2) / ((upper - lower) * (c - lower)),
                 sw(x < upper, 1 - ((upper - x) ** 2) / ((upper - lower) * (upper - c)), 1.0)))


@_wrap
def triangular_ppf(q, lower, c, upper):
    return sw(q < ((c - lower) / (upper - lower)), lower + np.sqrt((upper - lower) * (c - lower) * q),
              upper - np.sqrt((upper - lower) * (upper - c) * (1 - q)))


@_wrap
def triangular_isf(q, lower, c, upper):
    return triangular_ppf(1 - q, lower, c, upper)


@_wrap
def triangular_logpdf(x, lower, c, upper):
    res = sw(x < c, np.log(2 * (x - lower) / ((upper - lower) * (c - lower))),
             np.log(2 * (upper - x) / ((upper - lower) * (upper - c))))
    return sw((lower <= x) & (x <= upper), res, -inf)


@_wrap
def triangular_logcdf(x, lower, c, upper):
    return sw(x <= lower, -inf,
              sw(x <= c, np.log(((x - lower) ** 2) / ((upper - lower) * (c - lower))),
                 sw(x < upper, np.log1p(-((upper - x) ** 2) / ((upper - lower) * (upper - c))), 0)))


@_wrap
def triangular_logsf(x, lower, c, upper):
    return sw(x >= upper, -inf,
              sw(x >= c, np.log(((upper - x) ** 2) / ((upper - lower) * (upper - c))),
                 sw(x > lower, np.log1p(-((x - lower) ** 2) / ((upper - lower) * (c - lower))), 0)))


@_wrap
def triangular_sf(x, lower, c, upper):
    return np.exp(triangular_logsf(x, lower, c, upper))


# ------------------------------------------------------------------------------- kumaraswamy.py
@_wrap
def kumaraswamy_cdf(x, a, b):
    return cdf_bounds(1 - (1 - x**a) ** b, x, 0, 1)


@_wrap
def kumaraswamy_logpdf(x, a, b):
    return sw((x <= 0) | (x >= 1), -inf, np.log(a * b) + xlogy0(a - 1, x) + xlogy0(b - 1, 1 - x**a))


@_wrap
def kumaraswamy_pdf(x, a, b):
    return np.exp(kumaraswamy_logpdf(x, a, b))


@_wrap
def kumaraswamy_ppf(q, a, b):
    return ppf_bounds_cont((1 - (1 - q) ** (1 / b)) ** (1 / a), q, 0, 1)


@_wrap
def kumaraswamy_sf(x, a, b):
    return sf_bounds((1 - x**a) ** b, x, 0, 1)


@_wrap
def kumaraswamy_isf(x, a, b):
    return kumaraswamy_ppf(1 - x, a, b)


@_wrap
def kumaraswamy_logsf(x, a, b):
    return sw(x <= 0, 0.0, sw(x >= 1, -inf, b * np.log(1 - x**a)))


@_wrap
def kumaraswamy_logcdf(x, a, b):
    return sw(x <= 0, -inf, sw(x >= 1, 0.0, np.log(1 - (1 - x**a) ** b)))


# ------------------------------------------------------------------------------------- moyal.py
@_wrap
def moyal_cdf(x, mu, sigma):
    z = (x - mu) / sigma
    return 1 - sp.erf(np.exp(-z / 2) * (2**-0.5))


@_wrap
def moyal_logpdf(x, mu, sigma):
    z = (x - mu) / sigma
    return -(1 / 2) * (z + np.exp(-z)) - np.log(sigma) - (1 / 2) * np.log(2 * np.pi)


@_wrap
def moyal_pdf(x, mu, sigma):
    return np.exp(moyal_logpdf(x, mu, sigma))


@_wrap
def moyal_ppf(q, mu, sigma):
    return ppf_bounds_cont(sigma * -np.log(2.0 * sp.erfinv(1 - q) ** 2) + mu, q, -inf, inf)


@_wrap
def moyal_sf(x, mu, sigma):
    return 1 - moyal_cdf(x, mu, sigma)


@_wrap
def moyal_isf(x, mu, sigma):
    return moyal_ppf(1 - x, mu, sigma)


@_wrap
def moyal_logcdf(x, mu, sigma):
    z = (x - mu) / sigma
    return np.log(1 - sp.erf(np.exp(-z / 2) * (2**-0.5)))


@_wrap
def moyal_logsf(x, mu, sigma):
    return np.log(moyal_sf(x, mu, sigma))


# ------------------------------------------------------------------------- asymmetriclaplace.py
@_wrap
def asymmetriclaplace_cdf(x, mu, b, kappa):
    xn = (x - mu) / b
    kinv = 1 / kappa
    kk = kappa + kinv
    return sw(xn >= 0, 1 - np.exp(-xn * kappa) * (kinv / kk), np.exp(xn * kinv) * (kappa / kk))


@_wrap
def asymmetriclaplace_ppf(q, mu, b, kappa):
    kinv = 1 / kappa
    kk = kappa + kinv
    qp = sw(q >= kappa / kk, -np.log((1 - q) * kk * kappa) * kinv, np.log(q * kk / kappa) * kappa)
    return ppf_bounds_cont(qp * b + mu, q, -inf, inf)


@_wrap
def asymmetriclaplace_logpdf(x, mu, b, kappa):
    xn = (x - mu) / b
    kinv = 1 / kappa
    a = sw(xn >= 0, xn * -kappa, xn * kinv)
    a = a - np.log(kappa + kinv)
    return a - np.log(b)


@_wrap
def asymmetriclaplace_pdf(x, mu, b, kappa):
    return np.exp(asymmetriclaplace_logpdf(x, mu, b, kappa))


@_wrap
def asymmetriclaplace_sf(x, mu, b, kappa):
    return 1 - asymmetriclaplace_cdf(x, mu, b, kappa)


@_wrap
def asymmetriclaplace_isf(x, mu, b, kappa):
    return asymmetriclaplace_ppf(1 - x, mu, b, kappa)


@_wrap
def asymmetriclaplace_logcdf(x, mu, b, kappa):
    return np.log(asymmetriclaplace_cdf(x, mu, b, kappa))


@_wrap
def asymmetriclaplace_logsf(x, mu, b, kappa):
    return np.log(asymmetriclaplace_sf(x, mu, b, kappa))


# ------------------------------------------------------------------------------- loglogistic.py
@_wrap
def loglogistic_logpdf(x, alpha, beta):
    z = x / alpha
    res = np.log(beta) - np.log(alpha) + (beta - 1) * np.log(z) - 2 * np.log1p(z**beta)
    return sw(x <= 0, -inf, sw(np.isinf(x), -inf, res))


@_wrap
def loglogistic_pdf(x, alpha, beta):
    return np.exp(loglogistic_logpdf(x, alpha, beta))


@_wrap
def loglogistic_cdf(x, alpha, beta):
    z = x / alpha
    return cdf_bounds(1 / (1 + z ** (-beta)), x, 0, inf)


@_wrap
def loglogistic_logcdf(x, alpha, beta):
    z = x / alpha
    return sw(x <= 0, -inf, -np.log1p(z ** (-beta)))


@_wrap
def loglogistic_sf(x, alpha, beta):
    z = x / alpha
    return sw(x <= 0, 1.0, 1 / (1 + z**beta))


@_wrap
def loglogistic_logsf(x, alpha, beta):
    z = x / alpha
    return sw(x <= 0, 0.0, -np.log1p(z**beta))


@_wrap
def loglogistic_ppf(q, alpha, beta):
    return ppf_bounds_cont(alpha * (q / (1 - q)) ** (1 / beta), q, 0, inf)


@_wrap
def loglogistic_isf(q, alpha, beta):
    return loglogistic_ppf(1 - q, alpha, beta)


# ------------------------------------------------------------------------------- logitnormal.py
def _logit(x):
    return np.log(x) - np.log1p(-x)


@_wrap
def logitnormal_logpdf(x, mu, sigma):
    lx = _logit(x)
    return sw((x <= 0) | (x >= 1), -inf,
              -0.5 * ((lx - mu) / sigma) ** 2 - np.log(sigma) - 0.5 * np.log(2 * np.pi) - np.log(x) - np.log1p(-x))


@_wrap
def logitnormal_pdf(x, mu, sigma):
    return np.exp(logitnormal_logpdf(x, mu, sigma))


@_wrap
def logitnormal_cdf(x, mu, sigma):
    return cdf_bounds(0.5 * (1 + sp.erf((_logit(x) - mu) / (sigma * np.sqrt(2)))), x, 0, 1)


@_wrap
def logitnormal_logcdf(x, mu, sigma):
    z = (_logit(x) - mu) / sigma
    return sw(x <= 0, -inf, sw(x >= 1, 0.0, sw(z < -1.0, np.log(sp.erfcx(-z / np.sqrt(2.0)) / 2.0) - np.square(z) / 2.0,
                                               np.log1p(-sp.erfc(z / np.sqrt(2.0)) / 2.0))))


@_wrap
def logitnormal_logsf(x, mu, sigma):
    z = (_logit(x) - mu) / sigma
    return sw(x <= 0, 0.0, sw(x >= 1, -inf, sw(z > 1.0, np.log(sp.erfcx(z / np.sqrt(2.0)) / 2.0) - np.square(z) / 2.0,
                                               np.log1p(-0.5 * (1 + sp.erf(z / np.sqrt(2.0)))))))


@_wrap
def logitnormal_sf(x, mu, sigma):
    return np.exp(logitnormal_logsf(x, mu, sigma))


@_wrap
def logitnormal_ppf(q, mu, sigma):
    return ppf_bounds_cont(sigmoid(normal_ppf(q, mu, sigma)), q, 0, 1)


@_wrap
def logitnormal_isf(q, mu, sigma):
    return logitnormal_ppf(1 - q, mu, sigma)


# ------------------------------------------------------------------------------ inversegamma.py
@_wrap
def inversegamma_cdf(x, alpha, beta):
    return cdf_bounds(sp.gammaincc(alpha, beta / x), x, 0, inf)


@_wrap
def inversegamma_logpdf(x, alpha, beta):
    return sw(x <= 0, -inf, alpha * np.log(beta) - sp.gammaln(alpha) - (alpha + 1) * np.log(x) - beta / x)


@_wrap
def inversegamma_pdf(x, alpha, beta):
    return np.exp(inversegamma_logpdf(x, alpha, beta))


@_wrap
def inversegamma_ppf(q, alpha, beta):
    return ppf_bounds_cont(beta / sp.gammaincinv(alpha, 1 - q), q, 0, inf)


@_wrap
def inversegamma_isf(x, alpha, beta):
    return inversegamma_ppf(1 - x, alpha, beta)


@_wrap
def inversegamma_logcdf(x, alpha, beta):
    return sw(x <= 0, -inf, np.log(sp.gammaincc(alpha, beta / x)))


@_wrap
def inversegamma_logsf(x, alpha, beta):
    return sw(x <= 0, 0.0, np.log(sp.gammainc(alpha, beta / x)))


@_wrap
def inversegamma_sf(x, alpha, beta):
    return np.exp(inversegamma_logsf(x, alpha, beta))


# -------------------------------------------------------------------------------- chisquared.py
@_wrap
def chisquared_cdf(x, nu):
    return cdf_bounds(sp.gammainc(nu / 2, x / 2), x, 0, inf)


@_wrap
def chisquared_ppf(q, nu):
    return ppf_bounds_cont(2 * sp.gammaincinv(nu / 2, q), q, 0, inf)


@_wrap
def chisquared_logpdf(x, nu):
    return sw(x < 0, -inf, xlogy0(nu / 2 - 1, x) - x / 2 - sp.gammaln(nu / 2) - (nu * np.log(2)) / 2)


@_wrap
def chisquared_pdf(x, nu):
    return np.exp(chisquared_logpdf(x, nu))


@_wrap
def chisquared_logcdf(x, nu):
    return sw(x < 0.0, -inf, sw(x < nu, np.log(sp.gammainc(nu / 2, x / 2)), np.log1p(-sp.gammaincc(nu / 2, x / 2))))


@_wrap
def chisquared_sf(x, nu):
    return 1 - chisquared_cdf(x, nu)


@_wrap
def chisquared_isf(x, nu):
    return chisquared_ppf(1 - x, nu)


@_wrap
def chisquared_logsf(x, nu):
    return sw(x < 0.0, 0.0, sw(x < nu, np.log1p(-sp.gammainc(nu / 2, x / 2)), np.log(sp.gammaincc(nu / 2, x / 2))))


# ------------------------------------------------------------------- scaledinversechisquared.py
@_wrap
def scaledinversechisquared_cdf(x, nu, tau2):
    h = nu / 2
    return cdf_bounds(sp.gammaincc(h, h * tau2 / x), x, 0, inf)


@_wrap
def scaledinversechisquared_logpdf(x, nu, tau2):
    h = nu / 2
    return sw(x <= 0, -inf,
              -(np.log(x) * (h + 1)) - (h * tau2) / x + np.log(tau2) * h - sp.gammaln(h) + np.log(h) * h)


@_wrap
def scaledinversechisquared_pdf(x, nu, tau2):
    return np.exp(scaledinversechisquared_logpdf(x, nu, tau2))


@_wrap
def scaledinversechisquared_ppf(q, nu, tau2):
    h = nu / 2
    return ppf_bounds_cont(h * tau2 / sp.gammaincinv(h, 1 - q), q, 0, inf)


@_wrap
def scaledinversechisquared_isf(x, nu, tau2):
    return scaledinversechisquared_ppf(1 - x, nu, tau2)


@_wrap
def scaledinversechisquared_logcdf(x, nu, tau2):
    h = nu / 2
    return sw(x <= 0, -inf, np.log(sp.gammaincc(h, h * tau2 / x)))


@_wrap
def scaledinversechisquared_logsf(x, nu, tau2):
    h = nu / 2
    return sw(x <= 0, 0.0, np.log(sp.gammainc(h, h * tau2 / x)))


@_wrap
def scaledinversechisquared_sf(x, nu, tau2):
    return np.exp(scaledinversechisquared_logsf(x, nu, tau2))


# ------------------------------------------------------------------------------ halfstudentt.py
@_wrap
def halfstudentt_cdf(x, nu, sigma):
    xn = x / sigma
    factor = 0.5 * sp.betainc(0.5 * nu, 0.5, nu / (xn**2 + nu))
    c = sw(xn < 0, factor, 1 - factor) * 2 - 1
    return sw(nu > 1e5, halfnormal_cdf(x, sigma), cdf_bounds(c, x, 0, inf))


@_wrap
def halfstudentt_logpdf(x, nu, sigma):
    t = sw(x < 0, -inf,
           sp.gammaln((nu + 1) / 2) - sp.gammaln(nu / 2) - 0.5 * np.log(nu * np.pi * sigma**2)
           - 0.5 * (nu + 1) * np.log(1 + (x / sigma) ** 2 / nu) + np.log(2))
    return sw(nu > 1e5, halfnormal_logpdf(x, sigma), t)


@_wrap
def halfstudentt_pdf(x, nu, sigma):
    return np.exp(halfstudentt_logpdf(x, nu, sigma))


@_wrap
def halfstudentt_ppf(q, nu, sigma):
    qf = (q + 1) / 2
    v = sw(qf < 0.5, sp.betaincinv(0.5 * nu, 0.5, 2 * qf), np.sqrt(nu / sp.betaincinv(0.5 * nu, 0.5, 2 - 2 * qf) - nu))
    return ppf_bounds_cont(v * sigma, q, 0, inf)


@_wrap
def halfstudentt_isf(x, nu, sigma):
    return halfstudentt_ppf(1 - x, nu, sigma)


@_wrap
def halfstudentt_sf(x, nu, sigma):
    return 1 - halfstudentt_cdf(x, nu, sigma)


@_wrap
def halfstudentt_logcdf(x, nu, sigma):
    return np.log(halfstudentt_cdf(x, nu, sigma))


@_wrap
def halfstudentt_logsf(x, nu, sigma):
    return log1mexp(halfstudentt_logcdf(x, nu, sigma))


# -------------------------------------------------------------------------------- betascaled.py
@_wrap
def betascaled_logcdf(x, alpha, beta, lower, upper):
    xn = (x - lower) / (upper - lower)
    return sw(x < lower, -inf, sw(x > upper, 0.0, np.log(sp.betainc(alpha, beta, xn))))


@_wrap
def betascaled_cdf(x, alpha, beta, lower, upper):
    return np.exp(betascaled_logcdf(x, alpha, beta, lower, upper))


@_wrap
def betascaled_logpdf(x, alpha, beta, lower, upper):
    return sw((x < lower) | (x > upper), -inf,
              (xlogy0(alpha - 1, x - lower) + xlogy0(beta - 1, upper - x))
              - (xlogy0(alpha + beta - 1, upper - lower) + betaln(alpha, beta)))


@_wrap
def betascaled_pdf(x, alpha, beta, lower, upper):
    return np.exp(betascaled_logpdf(x, alpha, beta, lower, upper))


@_wrap
def betascaled_ppf(q, alpha, beta, lower, upper):
    return ppf_bounds_cont(sp.betaincinv(alpha, beta, q) * (upper - lower) + lower, q, lower, upper)


@_wrap
def betascaled_isf(x, alpha, beta, lower, upper):
    return betascaled_ppf(1 - x, alpha, beta, lower, upper)


@_wrap
def betascaled_sf(x, alpha, beta, lower, upper):
    return 1 - betascaled_cdf(x, alpha, beta, lower, upper)


@_wrap
def betascaled_logsf(x, alpha, beta, lower, upper):
    xn = (x - lower) / (upper - lower)
    return sw(x < lower, 0.0, sw(x > upper, -inf, np.log(sp.betainc(beta, alpha, 1 - xn))))


FAMILY_TABLE = {
    name: obj
    for name, obj in list(globals().items())
    if callable(obj) and "_" in name and not name.startswith("_") and name.split("_")[-1] in
    ("pdf", "logpdf", "cdf", "logcdf", "sf", "logsf", "ppf", "isf")
}
