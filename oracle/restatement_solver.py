"""CPU oracle for the solver families -- TEST INFRASTRUCTURE ONLY.

NumPy/SciPy float64 restatement, line by line, of the reference modules wald.py, exgaussian.py,
truncatednormal.py, skew_studentt.py, skewnormal.py, vonmises.py (+ helper.py:337-372), rice.py (+ helper.py:449-539), polyagamma.py and of `find_ppf` (optimization.py:10-32); citations are
file:line of /root/reference/pytensor_distributions.  Pinned against the reference's own SciPy
grids (tests/test_wald.py:10-19, test_exgaussian.py:10-18, test_truncatednormal.py:10-26,
test_skew_studentt.py:11-21) by tests/test_oracle_pinning.py.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from __future__ import annotations

import numpy as np
from scipy import special as sp

from .restatement import (_F, _f, _quiet, betaln, cdf_bounds, log1mexp, logdiffexp, normal_logcdf, normal_logpdf,
                          ppf_bounds_cont)

inf = np.inf
sw = np.where
_SQRT2 = np.sqrt(2.0)


def _wrap(fn):
    def g(*args):
        with _quiet():
            return np.asarray(fn(*_f(*args)), dtype=_F)

    g.__name__ = fn.__name__
    return g


def find_ppf(q, x0, lower, upper, cdf_func, pdf_func, *params, max_iter=100, tol=1e-8):
    """optimization.py:10-32: clipped Newton on the whole batch until every element has converged;
    a converged element keeps x_prev (and so converges again in every later iteration)."""
    q = np.asarray(q, dtype=_F)
    x = x0 + np.zeros_like(q)
    with _quiet():
        for _ in range(max_iter):
            cdf_val = cdf_func(x, *params)
            f_x = np.maximum(pdf_func(x, *params), 1e-10)
            delta = (cdf_val - q) / f_x
            max_step = np.maximum(np.abs(x), 1.0)
            delta = sw(delta < -max_step, -max_step, sw(delta > max_step, max_step, delta))  # pt.clip keeps NaN
            x_new = x - delta
            converged = np.abs(x_new - x) < tol
            x = sw(converged, x, x_new)
            if np.all(converged):
                break
    return ppf_bounds_cont(x, q, lower, upper)


# -------------------------------------------------------------------------------------- wald.py
@_wrap
def wald_logpdf(x, mu, lam):  # wald.py:77-82
    r = 0.5 * (np.log(lam) - np.log(2 * np.pi) - 3 * np.log(x) - lam * (x - mu) ** 2 / (mu**2 * x))
    return sw((x <= 0) | (x == inf), -inf, r)


@_wrap
def wald_pdf(x, mu, lam):  # wald.py:54-55
    return np.exp(wald_logpdf(x, mu, lam))


@_wrap
def wald_cdf(x, mu, lam):  # wald.py:40-47
    eps = 1e-12
    u = np.sqrt(lam / (x + eps))
    v = x / mu
    z1 = 0.5 * (1 + sp.erf(u * (v - 1) / _SQRT2))
    z2 = np.exp(2 * lam / mu) * 0.5 * (1 + sp.erf(-u * (v + 1) / _SQRT2))
    return cdf_bounds(z1 + z2, x, 0, inf)


def _wald_mode(mu, lam):  # wald.py:12-14
    ratio = 3 * mu / (2 * lam)
    return mu * (np.sqrt(1 + ratio**2) - ratio)


@_wrap
def wald_ppf(q, mu, lam):  # wald.py:58-60
    return find_ppf(q, _wald_mode(mu, lam), 0, inf, wald_cdf, wald_pdf, mu, lam)


@_wrap
def wald_isf(q, mu, lam):  # wald.py:50-51
    return wald_ppf(1 - q, mu, lam)


@_wrap
def wald_logcdf(x, mu, lam):  # wald.py:73-74
    return np.log(wald_cdf(x, mu, lam))


@_wrap
def wald_logsf(x, mu, lam):  # wald.py:85-86
    return np.log1p(-wald_cdf(x, mu, lam))


@_wrap
def wald_sf(x, mu, lam):  # wald.py:63-64
    return np.exp(wald_logsf(x, mu, lam))


# -------------------------------------------------------------------------------- exgaussian.py
@_wrap
def exgaussian_logpdf(x, mu, sigma, nu):  # exgaussian.py:101-111
    t = -np.log(nu) + (mu - x) / nu + 0.5 * (sigma / nu) ** 2 + normal_logcdf(x - (mu + sigma**2 / nu), 0.0, sigma)
    return sw(nu >= 0.05 * sigma, t, normal_logpdf(x, mu, sigma))


@_wrap
def exgaussian_logcdf(x, mu, sigma, nu):  # exgaussian.py:76-98
    a = normal_logcdf(x, mu, sigma)
    b = (mu - x) / nu + 0.5 * (sigma / nu) ** 2 + normal_logcdf(x - (mu + sigma**2 / nu), 0.0, sigma)
    r = sw(nu >= 0.05 * sigma, logdiffexp(a, b), a)
    return sw(x == -inf, -inf, sw(x == inf, 0.0, r))


@_wrap
def exgaussian_cdf(x, mu, sigma, nu):  # exgaussian.py:50-51
    return np.exp(exgaussian_logcdf(x, mu, sigma, nu))


@_wrap
def exgaussian_pdf(x, mu, sigma, nu):  # exgaussian.py:58-59
    return np.exp(exgaussian_logpdf(x, mu, sigma, nu))


@_wrap
def exgaussian_sf(x, mu, sigma, nu):  # exgaussian.py:66-67
    return 1.0 - exgaussian_cdf(x, mu, sigma, nu)


@_wrap
def exgaussian_logsf(x, mu, sigma, nu):  # exgaussian.py:114-115
    return np.log1p(-np.exp(exgaussian_logcdf(x, mu, sigma, nu)))


@_wrap
def exgaussian_ppf(q, mu, sigma, nu):  # exgaussian.py:62-63 (x0 = mean = mu + nu)
    return find_ppf(q, mu + nu, -inf, inf, exgaussian_cdf, exgaussian_pdf, mu, sigma, nu)


@_wrap
def exgaussian_isf(q, mu, sigma, nu):  # exgaussian.py:54-55
    return exgaussian_ppf(1 - q, mu, sigma, nu)


# --------------------------------------------------------------------------- truncatednormal.py
def _phi(z):  # truncatednormal.py:7-8
    return 0.5 * (1 + sp.erf(z / _SQRT2))


def _alpha_beta(mu, sigma, lower, upper):  # truncatednormal.py:19-25
    a_raw = (lower - mu) / sigma
    b_raw = (upper - mu) / sigma
    alpha = sw(np.isinf(a_raw) & (a_raw < 0), -100.0, a_raw)
    beta = sw(np.isinf(b_raw) & (b_raw > 0), 100.0, b_raw)
    return alpha, beta


def _log_erfc(x):  # truncatednormal.py:28-31
    return np.log(sp.erfcx(x)) - x**2


def _log_Z(alpha, beta):  # truncatednormal.py:34-61
    a, b = alpha / _SQRT2, beta / _SQRT2
    log_half = np.log(0.5)
    sa, sb = sw(alpha > 0, a, 0.0), sw(alpha > 0, b, 0.0)
    res_pos = log_half + logdiffexp(_log_erfc(sa), _log_erfc(sb))
    nb, na = sw(beta < 0, -b, 0.0), sw(beta < 0, -a, 0.0)
    res_neg = log_half + logdiffexp(_log_erfc(nb), _log_erfc(na))
    res_straddle = log_half + np.log(sp.erf(b) - sp.erf(a))
    return sw(alpha > 0, res_pos, sw(beta < 0, res_neg, res_straddle))


@_wrap
def truncatednormal_logpdf(x, mu, sigma, lower, upper):  # truncatednormal.py:172-183
    alpha, beta = _alpha_beta(mu, sigma, lower, upper)
    return sw((x < lower) | (x > upper), -inf, normal_logpdf(x, mu, sigma) - _log_Z(alpha, beta))


@_wrap
def truncatednormal_pdf(x, mu, sigma, lower, upper):  # truncatednormal.py:168-169
    return np.exp(truncatednormal_logpdf(x, mu, sigma, lower, upper))


@_wrap
def truncatednormal_cdf(x, mu, sigma, lower, upper):  # truncatednormal.py:186-206
    alpha, beta = _alpha_beta(mu, sigma, lower, upper)
    xi = (x - mu) / sigma
    r = sw(xi < alpha, 0.0, sw(xi > beta, 1.0, np.exp(_log_Z(alpha, xi) - _log_Z(alpha, beta))))
    return sw(x <= lower, 0.0, sw(x >= upper, 1.0, r))


@_wrap
def truncatednormal_logcdf(x, mu, sigma, lower, upper):  # truncatednormal.py:209-222
    alpha, beta = _alpha_beta(mu, sigma, lower, upper)
    xi = (x - mu) / sigma
    return sw(x <= lower, -inf, sw(x >= upper, 0.0, _log_Z(alpha, xi) - _log_Z(alpha, beta)))


@_wrap
def truncatednormal_sf(x, mu, sigma, lower, upper):  # truncatednormal.py:225-226
    return 1.0 - truncatednormal_cdf(x, mu, sigma, lower, upper)


@_wrap
def truncatednormal_logsf(x, mu, sigma, lower, upper):  # truncatednormal.py:229-242
    alpha, beta = _alpha_beta(mu, sigma, lower, upper)
    xi = (x - mu) / sigma
    return sw(x <= lower, 0.0, sw(x >= upper, -inf, _log_Z(xi, beta) - _log_Z(alpha, beta)))


@_wrap
def truncatednormal_ppf(q, mu, sigma, lower, upper):  # truncatednormal.py:245-266
    alpha, beta = _alpha_beta(mu, sigma, lower, upper)
    Z = _phi(beta) - _phi(alpha)
    standard = _SQRT2 * sp.erfinv(2 * (q * Z + _phi(alpha)) - 1)
    sb = 0.5 * sp.erfc(beta / _SQRT2)
    sa = 0.5 * sp.erfc(alpha / _SQRT2)
    survival = _SQRT2 * sp.erfcinv(2 * (q * sb + (1 - q) * sa))
    r = mu + sigma * sw(alpha > 0, survival, standard)
    return ppf_bounds_cont(r, q, lower, upper)


@_wrap
def truncatednormal_isf(q, mu, sigma, lower, upper):  # truncatednormal.py:269-270
    return truncatednormal_ppf(1.0 - q, mu, sigma, lower, upper)


# ----------------------------------------------------------------------------- skew_studentt.py
@_wrap
def skew_studentt_logpdf(x, a, b, mu, sigma):  # skew_studentt.py:113-120
    z = (x - mu) / sigma
    s = np.sqrt(a + b + z**2)
    r = ((a + 0.5) * np.log(1 + z / s) + (b + 0.5) * np.log(1 - z / s)
         - ((a + b - 1) * np.log(2) + betaln(a, b) + 0.5 * np.log(a + b)) - np.log(sigma))
    return sw(np.isinf(x), np.nan, r)


@_wrap
def skew_studentt_pdf(x, a, b, mu, sigma):  # skew_studentt.py:109-110
    return np.exp(skew_studentt_logpdf(x, a, b, mu, sigma))


@_wrap
def skew_studentt_cdf(x, a, b, mu, sigma):  # skew_studentt.py:123-129
    z = (x - mu) / sigma
    y = 0.5 * (1 + z / np.sqrt(a + b + z**2))
    r = sp.betainc(a, b, y)
    return sw(x == inf, 1.0, sw(x == -inf, 0.0, r))


@_wrap
def skew_studentt_logcdf(x, a, b, mu, sigma):  # skew_studentt.py:132-138
    z = (x - mu) / sigma
    y = 0.5 * (1 + z / np.sqrt(a + b + z**2))
    r = np.log(sp.betainc(a, b, y))
    return sw(x == inf, 0.0, sw(x == -inf, -inf, r))


@_wrap
def skew_studentt_sf(x, a, b, mu, sigma):  # skew_studentt.py:141-143
    return skew_studentt_cdf(-x, b, a, -mu, sigma)


@_wrap
def skew_studentt_logsf(x, a, b, mu, sigma):  # skew_studentt.py:146-147
    return skew_studentt_logcdf(-x, b, a, -mu, sigma)


@_wrap
def skew_studentt_ppf(q, a, b, mu, sigma):  # skew_studentt.py:150-157
    bval = sp.betaincinv(a, b, q)
    z = ((2 * bval - 1) * np.sqrt(a + b)) / (2 * np.sqrt(bval * (1 - bval)))
    return ppf_bounds_cont(mu + sigma * z, q, -inf, inf)


@_wrap
def skew_studentt_isf(q, a, b, mu, sigma):  # skew_studentt.py:160-161
    return skew_studentt_ppf(1 - q, a, b, mu, sigma)


# -------------------------------------------------------------------------------- skewnormal.py
def _skewnormal_mean(mu, sigma, alpha):  # skewnormal.py:8-10
    return mu + sigma * np.sqrt(2 / np.pi) * alpha / np.sqrt(1 + alpha**2)


@_wrap
def skewnormal_cdf(x, mu, sigma, alpha):  # skewnormal.py:61-62
    return 0.5 * (1 + sp.erf((x - mu) / (sigma * 2**0.5))) - 2 * sp.owens_t((x - mu) / sigma, alpha)


@_wrap
def skewnormal_logcdf(x, mu, sigma, alpha):  # skewnormal.py:65-66
    return np.log(skewnormal_cdf(x, mu, sigma, alpha))


@_wrap
def skewnormal_logpdf(x, mu, sigma, alpha):  # skewnormal.py:95-100
    tau = 1.0 / (sigma**2)
    return (np.log(1 + sp.erf(((x - mu) * np.sqrt(tau) * alpha) / np.sqrt(2)))
            + (-tau * (x - mu) ** 2 + np.log(tau / np.pi / 2.0)) / 2.0)


@_wrap
def skewnormal_pdf(x, mu, sigma, alpha):  # skewnormal.py:73-74
    return np.exp(skewnormal_logpdf(x, mu, sigma, alpha))


@_wrap
def skewnormal_sf(x, mu, sigma, alpha):  # skewnormal.py:83-84
    return 1 - skewnormal_cdf(x, mu, sigma, alpha)


@_wrap
def skewnormal_logsf(x, mu, sigma, alpha):  # skewnormal.py:103-104
    return log1mexp(skewnormal_logcdf(x, mu, sigma, alpha))


@_wrap
def skewnormal_ppf(q, mu, sigma, alpha):  # skewnormal.py:77-80
    q, mu, sigma, alpha = np.broadcast_arrays(q, mu, sigma, alpha)
    return find_ppf(q, _skewnormal_mean(mu, sigma, alpha), -inf, inf, skewnormal_cdf, skewnormal_pdf, mu, sigma, alpha)


@_wrap
def skewnormal_isf(q, mu, sigma, alpha):  # skewnormal.py:69-70
    return skewnormal_ppf(1 - q, mu, sigma, alpha)


# ---------------------------------------------------------------------------------- vonmises.py
def von_mises_cdf(x, mu, kappa):
    """helper.py:337-372 (series in ive ratios for kappa < 50, normal approximation above)."""
    x, mu, kappa = np.broadcast_arrays(*_f(x, mu, kappa))
    xc = x - mu
    ix = np.round(xc / (2 * np.pi))
    xw = xc - ix * 2 * np.pi
    p = np.clip(np.round(1 + 28.0 + 0.5 * kappa - 100.0 / (kappa + 5.0)), 5, 50).astype(np.int32)
    n = np.arange(1, 51, dtype=_F).reshape((-1,) + (1,) * x.ndim)
    with _quiet():
        ratio = sp.ive(n, kappa) / sp.ive(0, kappa)
        terms = ratio * np.sin(n * xw) / n
        series = np.sum(sw(n <= p, terms, 0.0), axis=0)
        cdf_series = np.clip(0.5 + xw / (2 * np.pi) + series / np.pi, 0.0, 1.0)
        b = np.sqrt(2 / np.pi) / sp.ive(0.0, kappa)
        z = b * np.sin(xw / 2.0)
        cdf_norm = 0.5 * (1.0 + sp.erf(z / np.sqrt(2.0)))
    return sw(kappa < 50.0, cdf_series, cdf_norm) + ix


@_wrap
def vonmises_logpdf(x, mu, kappa):  # vonmises.py:45-46
    return kappa * (np.cos(x - mu) - 1) - np.log(2 * np.pi) - np.log(sp.ive(0, kappa))


@_wrap
def vonmises_pdf(x, mu, kappa):  # vonmises.py:41-42
    return np.exp(vonmises_logpdf(x, mu, kappa))


@_wrap
def vonmises_cdf(x, mu, kappa):  # vonmises.py:53-54
    return von_mises_cdf(x, mu, kappa)


@_wrap
def vonmises_logcdf(x, mu, kappa):  # vonmises.py:57-58
    return np.log(von_mises_cdf(x, mu, kappa))


@_wrap
def vonmises_sf(x, mu, kappa):  # vonmises.py:61-62
    return 1.0 - von_mises_cdf(x, mu, kappa)


@_wrap
def vonmises_logsf(x, mu, kappa):  # vonmises.py:65-66
    return np.log1p(-von_mises_cdf(x, mu, kappa))


@_wrap
def vonmises_ppf(q, mu, kappa):  # vonmises.py:73-74 (x0 = mean = mu)
    q, mu, kappa = np.broadcast_arrays(q, mu, kappa)
    return find_ppf(q, mu, -inf, inf, von_mises_cdf, vonmises_pdf, mu, kappa)


@_wrap
def vonmises_isf(q, mu, kappa):  # vonmises.py:69-70
    return vonmises_ppf(1 - q, mu, kappa)


# -------------------------------------------------------------------------------------- rice.py
def ncx2_cdf(x, df, nc):
    """helper.py:449-512: 800-term Poisson-weighted series of gammainc for nc < 1000, normal approximation above."""
    x, nc = np.broadcast_arrays(*_f(x, nc))
    half_nc, half_df, half_x = nc / 2.0, df / 2.0, x / 2.0
    with _quiet():
        log_half_nc = sw(half_nc > 0, np.log(half_nc), -inf)
        series = np.zeros(x.shape)
        for lo in range(0, 800, 100):  # the reference builds the (800, ...) array at once; summed in order here
            j = np.arange(lo, lo + 100, dtype=_F).reshape((-1,) + (1,) * x.ndim)
            log_w = -half_nc + j * log_half_nc - sp.gammaln(j + 1)
            w = np.exp(np.clip(log_w, -700, 700))
            series = series + np.sum(w * sp.gammainc(half_df + j, half_x), axis=0)
        mean_approx = df + nc
        std_approx = np.sqrt(2 * (df + 2 * nc))
        normal_result = 0.5 * (1 + sp.erf((x - mean_approx) / (std_approx * np.sqrt(2))))
    return sw(nc < 1000, series, normal_result)


def marcum_q1_complement(a, b):  # helper.py:515-539
    a, b = _f(a, b)
    return sw(b <= 0, 0.0, ncx2_cdf(b**2, 2.0, a**2))


def _laguerre_half(q):  # rice.py:13-22
    return (1 + q) * sp.ive(0, q / 2) + q * sp.ive(1, q / 2)


def _rice_mean(nu, sigma):  # rice.py:25-28
    return sigma * np.sqrt(np.pi / 2) * _laguerre_half(nu**2 / (2 * sigma**2))


@_wrap
def rice_logpdf(x, nu, sigma):  # rice.py:65-78
    sigma2 = sigma**2
    z = x * nu / sigma2
    r = np.log(x) - 2 * np.log(sigma) - (x**2 + nu**2) / (2 * sigma2) + np.log(sp.ive(0, z)) + z
    return sw(x <= 0, -inf, sw(np.isinf(x), np.nan, r))


@_wrap
def rice_pdf(x, nu, sigma):  # rice.py:61-62
    return np.exp(rice_logpdf(x, nu, sigma))


@_wrap
def rice_cdf(x, nu, sigma):  # rice.py:81-83
    return cdf_bounds(marcum_q1_complement(nu / sigma, x / sigma), x, 0, inf)


@_wrap
def rice_logcdf(x, nu, sigma):  # rice.py:86-91
    return sw(x <= 0, -inf, np.log(rice_cdf(x, nu, sigma)))


@_wrap
def rice_sf(x, nu, sigma):  # rice.py:94-95
    return 1.0 - rice_cdf(x, nu, sigma)


@_wrap
def rice_logsf(x, nu, sigma):  # rice.py:98-99
    return np.log(rice_sf(x, nu, sigma))


@_wrap
def rice_ppf(q, nu, sigma):  # rice.py:102-103
    q, nu, sigma = np.broadcast_arrays(q, nu, sigma)
    return find_ppf(q, _rice_mean(nu, sigma), 0, inf, rice_cdf, rice_pdf, nu, sigma)


@_wrap
def rice_isf(q, nu, sigma):  # rice.py:106-107
    return rice_ppf(1.0 - q, nu, sigma)


# -------------------------------------------------------------------------------- polyagamma.py
_PG_LOWER = 1e-10  # polyagamma.py:15-16


def _pg_log_cosh_half(z):  # polyagamma.py:25-28
    a = np.abs(z / 2)
    return a + np.log1p(np.exp(-2 * a)) - np.log(2.0)


def _pg_log_density_base(x, h, N=20):  # polyagamma.py:31-68
    n = np.arange(N, dtype=_F).reshape((-1,) + (1,) * np.ndim(x))
    c = 2 * n + h
    signs = sw(n % 2 == 0, 1.0, -1.0)
    log_abs = (sp.gammaln(n + h) - sp.gammaln(n + 1) + np.log(c) - 0.5 * np.log(2 * np.pi) - 1.5 * np.log(x)
               - c**2 / (8 * x))
    max_log = np.max(log_abs, axis=0)
    signed_sum = np.sum(signs * np.exp(log_abs - max_log), axis=0)
    log_series = np.log(np.maximum(signed_sum, 1e-300)) + max_log
    return (h - 1) * np.log(2.0) - sp.gammaln(h) + log_series


def _pg_mean(h, z):  # polyagamma.py:71-77
    small = np.abs(z) < 1e-6
    sz = sw(small, 1.0, z)
    return sw(small, h / 4.0, h / (2 * sz) * np.tanh(sz / 2))


def _pg_var(h, z):  # polyagamma.py:88-94
    small = np.abs(z) < 1e-6
    sz = sw(small, 1.0, z)
    return sw(small, h / 24.0, h / (4 * sz**3) * (np.sinh(sz) - sz) / np.cosh(sz / 2) ** 2)


@_wrap
def polyagamma_logpdf(x, h, z):  # polyagamma.py:113-117
    x, h, z = np.broadcast_arrays(x, h, z)
    r = h * _pg_log_cosh_half(z) - z**2 * x / 2 + _pg_log_density_base(x, h)
    return sw(x <= 0, -inf, r)


@_wrap
def polyagamma_pdf(x, h, z):  # polyagamma.py:120-121
    return np.exp(polyagamma_logpdf(x, h, z))


@_wrap
def polyagamma_cdf(x, h, z):  # polyagamma.py:124-134: 500-point trapezoid of the pdf from 1e-10 to x
    x, h, z = np.broadcast_arrays(x, h, z)
    n_points = 500
    step = (x - _PG_LOWER) / (n_points - 1)
    i = np.arange(n_points, dtype=_F).reshape((-1,) + (1,) * x.ndim)
    t = _PG_LOWER + i * step  # pt.linspace
    pv = polyagamma_pdf(t, h, z)
    result = step * (0.5 * pv[0] + np.sum(pv[1:-1], axis=0) + 0.5 * pv[-1])
    return cdf_bounds(result, x, 0, inf)


@_wrap
def polyagamma_logcdf(x, h, z):  # polyagamma.py:137-138
    return sw(x <= 0, -inf, np.log(polyagamma_cdf(x, h, z)))


@_wrap
def polyagamma_sf(x, h, z):  # polyagamma.py:141-142
    return 1.0 - polyagamma_cdf(x, h, z)


@_wrap
def polyagamma_logsf(x, h, z):  # polyagamma.py:145-146
    return np.log1p(-polyagamma_cdf(x, h, z))


@_wrap
def polyagamma_ppf(q, h, z, max_iter=50, tol=1e-8):  # polyagamma.py:149-184
    q, h, z = np.broadcast_arrays(q, h, z)
    m, v = _pg_mean(h, z), _pg_var(h, z)
    s2 = np.log1p(v / m**2)
    mu_ln = np.log(m) - s2 / 2
    x = np.exp(mu_ln + np.sqrt(s2) * np.sqrt(2.0) * sp.erfinv(2 * q - 1))
    x = np.maximum(x, _PG_LOWER)
    for _ in range(max_iter):
        delta = (polyagamma_cdf(x, h, z) - q) / np.maximum(polyagamma_pdf(x, h, z), 1e-10)
        ms = np.maximum(np.abs(x), 0.5)
        delta = sw(delta < -ms, -ms, sw(delta > ms, ms, delta))
        x_new = np.maximum(x - delta, _PG_LOWER)
        converged = np.abs(x_new - x) < tol
        x = sw(converged, x, x_new)
        if np.all(converged):
            break
    return ppf_bounds_cont(x, q, 0, inf)


@_wrap
def polyagamma_isf(q, h, z):  # polyagamma.py:187-188
    return polyagamma_ppf(1.0 - q, h, z)


SOLVER_TABLE = {k: v for k, v in list(globals().items())
                if callable(v) and k.split("_")[-1] in ("pdf", "logpdf", "cdf", "logcdf", "sf", "logsf", "ppf", "isf")
                and not k.startswith(("normal_", "_"))}
