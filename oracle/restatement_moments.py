"""TEST INFRASTRUCTURE ONLY (see oracle/restatement.py): NumPy / SciPy float64 restatement of the reference's
moment functions -- mean, mode, median, var, std, skewness, kurtosis, entropy of every module -- and of the
quadrature helpers they use (helper.py:99-322).  Each entry cites the reference lines it
follows.  pt.X -> scipy.special.X, pt.switch / pt.where -> np.where, `ppf`/`logpdf`/`pdf` are the restated
functions of oracle/restatement.py.  Parameters are evaluated per parameter set (the reference's helpers take
`arange(min, max)` of the parameters, which is defined for one parameter set at a time; discrete_moment's range is
the min / max of the quantiles over the batch, helper.py:203-204, i.e. the own quantiles for one parameter set).

Pinned by the reference's own tests: tests/helper_scipy.py:45-48,164-238 compare every function with scipy.stats at
rtol 1e-6 (entropy 1e-4) on the parameter grids of tests/test_<dist>.py -- tests/test_oracle_moments.py repeats
exactly that on the restatement.
"""

import numpy as np
import scipy.special as sp

from . import restatement as R

STATS = ("mean", "mode", "median", "var", "std", "skewness", "kurtosis", "entropy")
EULER = 0.5772156649015329  # pt.euler_gamma


def _f(*ps):
    return [np.asarray(p, dtype=np.float64) for p in ps]


def _full(v, *ps):
    """pt.full_like(pt.broadcast_arrays(*params)[0], v)"""
    shape = np.broadcast_shapes(*[np.shape(p) for p in ps], np.shape(v))
    return np.broadcast_to(np.asarray(v, dtype=np.float64), shape).copy()


def _fn(name, fn):
    return R.TABLE[f"{name}_{fn}"]


# ------------------------------------------------------------------ helper.py:99-322 (per parameter set)
def _per_set(f, *ps):
    ps = np.broadcast_arrays(*_f(*ps))
    out = np.empty(ps[0].shape)
    it = np.nditer(ps[0], flags=["multi_index"])
    for _ in it:
        out[it.multi_index] = f(*[float(p[it.multi_index]) for p in ps])
    return out if out.ndim else out[()]


def continuous_entropy(min_x, max_x, logpdf, *params, n_points=1000):
    """helper.py:99-142, one parameter set."""
    x = np.linspace(min_x, max_x, n_points)
    with np.errstate(all="ignore"):
        lp = logpdf(x, *params)
        integrand = -np.exp(lp) * lp
    dx = (max_x - min_x) / (n_points - 1)
    return dx * (0.5 * integrand[0] + np.sum(integrand[1:-1]) + 0.5 * integrand[-1])


def discrete_entropy(min_x, max_x, logpdf, *params):
    """helper.py:145-176, one parameter set; pt.arange(min_x, max_x)."""
    k = np.arange(min_x, max_x).astype(np.float64)
    with np.errstate(all="ignore"):
        lp = logpdf(k, *params)
        return float(np.sum(-np.exp(lp) * lp))


def discrete_moment(ppf, pdf, *params, order=1):
    """helper.py:179-214, one parameter set."""
    min_x = float(ppf(np.float64(0.0001), *params))
    max_x = float(ppf(np.float64(0.9999), *params)) + 1
    k = np.arange(min_x, max_x).astype(np.float64)
    p = pdf(k, *params)
    result = np.sum(k * p)
    if order > 1:
        result = np.sum((k - result) ** order * p)
    return float(result)


def continuous_moment(lower, upper, logpdf, *params, order=1, mean_val=None, n_points=1000):
    """helper.py:242-295, one parameter set."""
    x = np.linspace(lower, upper, n_points)
    with np.errstate(all="ignore"):
        p = np.exp(logpdf(x, *params))
        integrand = (x - mean_val) ** order * p if mean_val is not None else x**order * p
    dx = (upper - lower) / (n_points - 1)
    return dx * (0.5 * integrand[0] + np.sum(integrand[1:-1]) + 0.5 * integrand[-1])


def continuous_skewness(lower, upper, logpdf, *params):
    """helper.py:309-314"""
    m = continuous_moment(lower, upper, logpdf, *params, order=1)
    v = continuous_moment(lower, upper, logpdf, *params, order=2, mean_val=m)
    t = continuous_moment(lower, upper, logpdf, *params, order=3, mean_val=m)
    return t / (np.sqrt(v) ** 3)


def continuous_kurtosis(lower, upper, logpdf, *params):
    """helper.py:317-322"""
    m = continuous_moment(lower, upper, logpdf, *params, order=1)
    v = continuous_moment(lower, upper, logpdf, *params, order=2, mean_val=m)
    t = continuous_moment(lower, upper, logpdf, *params, order=4, mean_val=m)
    return t / (v**2) - 3


MOMENTS = {}


def _reg(name, **fns):
    assert set(fns) == set(STATS), (name, set(STATS) ^ set(fns))
    MOMENTS[name] = fns


def moment(name, stat, *params):
    with np.errstate(all="ignore"):
        return np.asarray(MOMENTS[name][stat](*_f(*params)), dtype=np.float64)


# ------------------------------------------------------------------------------------ normal.py:6-43
_reg("normal",
     mean=lambda mu, s: _full(mu, mu, s), mode=lambda mu, s: _full(mu, mu, s), median=lambda mu, s: _full(mu, mu, s),
     var=lambda mu, s: _full(np.square(s), mu, s), std=lambda mu, s: _full(s, mu, s),
     skewness=lambda mu, s: _full(0.0, mu, s), kurtosis=lambda mu, s: _full(0.0, mu, s),
     entropy=lambda mu, s: _full(0.5 * np.log(2 * np.pi * np.e * s**2), mu, s))

# ------------------------------------------------------------------------------------- gamma.py:8-37
_reg("gamma",
     mean=lambda a, b: a / b, mode=lambda a, b: np.where(a <= 1, 0.0, (a - 1) / b),
     median=lambda a, b: _fn("gamma", "ppf")(0.5, a, b),
     var=lambda a, b: a / np.square(b), std=lambda a, b: np.sqrt(a) / b,
     skewness=lambda a, b: _full(2 / np.sqrt(a), a, b), kurtosis=lambda a, b: _full(6 / a, a, b),
     entropy=lambda a, b: a - np.log(b) + sp.gammaln(a) + (1 - a) * sp.digamma(a))


# -------------------------------------------------------------------------------------- beta.py:9-63
def _beta_mode(a, b):
    a, b = np.broadcast_arrays(a, b)
    r = np.full(a.shape, np.nan)
    r = np.where((a == 1) & (b > 1), 0.0, r)
    r = np.where((b == 1) & (a > 1), 1.0, r)
    return np.where((a > 1) & (b > 1), (a - 1) / (a + b - 2), r)


def _beta_var(a, b):
    return (a * b) / (np.power(a + b, 2) * (a + b + 1))


def _beta_kurt(a, b):
    psc, prod = a + b, a * b
    return 6 * (np.abs(a - b) ** 2 * (psc + 1) - prod * (psc + 2)) / (prod * (psc + 2) * (psc + 3))


_reg("beta",
     mean=lambda a, b: a / (a + b), mode=_beta_mode, median=lambda a, b: _fn("beta", "ppf")(0.5, a, b),
     var=_beta_var, std=lambda a, b: np.sqrt(_beta_var(a, b)),
     skewness=lambda a, b: np.where(a == b, 0.0, (2 * (b - a) * np.sqrt(a + b + 1)) / ((a + b + 2) * np.sqrt(a * b))),
     kurtosis=_beta_kurt,
     entropy=lambda a, b: R.betaln(a, b) - (a - 1) * sp.psi(a) - (b - 1) * sp.psi(b) + (a + b - 2) * sp.psi(a + b))


# ---------------------------------------------------------------------------------- studentt.py:11-64
def _t_var(nu, mu, s):
    nu, mu, s = np.broadcast_arrays(nu, mu, s)
    r = np.full(nu.shape, np.nan)
    r = np.where(nu > 2, s**2 * nu / (nu - 2), r)
    r = np.where((nu > 1) & (nu <= 2), np.inf, r)
    return np.where(nu > 1e10, s**2, r)


def _t_std(nu, mu, s):
    v = _t_var(nu, mu, s)
    return np.where(np.isinf(v) | np.isnan(v), v, np.sqrt(v))


def _t_kurt(nu, mu, s):
    nu = np.broadcast_arrays(nu, mu, s)[0]
    r = np.full(nu.shape, np.nan)
    r = np.where(nu > 4, 6 / (nu - 4), r)
    return np.where((nu > 2) & (nu <= 4), np.inf, r)


def _t_entropy(nu, mu, s):
    normal = _full(0.5 * np.log(2 * np.pi * np.e * s**2), mu, s)
    t = np.log(s) + 0.5 * (nu + 1) * (sp.psi(0.5 * (nu + 1)) - sp.psi(0.5 * nu)) + np.log(np.sqrt(nu)) + R.betaln(0.5 * nu, 0.5)
    return np.where(nu > 1e5, normal, t)


_reg("studentt",
     mean=lambda nu, mu, s: np.where(_full(nu, nu, mu, s) > 1, _full(mu, nu, mu, s), np.nan),
     mode=lambda nu, mu, s: _full(mu, nu, mu, s), median=lambda nu, mu, s: _full(mu, nu, mu, s),
     var=_t_var, std=_t_std, skewness=lambda nu, mu, s: np.where(_full(nu, nu, mu, s) > 3, 0.0, np.nan),
     kurtosis=_t_kurt, entropy=_t_entropy)

# ------------------------------------------------------------------------------------ poisson.py:9-43
_reg("poisson",
     mean=lambda mu: mu * 1.0, mode=np.floor, median=np.floor, var=lambda mu: mu * 1.0, std=np.sqrt,
     skewness=lambda mu: 1.0 / np.sqrt(mu), kurtosis=lambda mu: 1.0 / mu,
     entropy=lambda mu: np.where(mu < 10, _per_set(lambda m: discrete_entropy(0, 25, _fn("poisson", "logpdf"), m), mu),
                                 0.5 * np.log(2.0 * np.pi * np.e * mu) - 1.0 / (12.0 * mu)))

# ----------------------------------------------------------------------------------- binomial.py:9-38
_reg("binomial",
     mean=lambda n, p: n * p, mode=lambda n, p: np.floor((n + 1) * p), median=lambda n, p: _fn("binomial", "ppf")(0.5, n, p),
     var=lambda n, p: n * p * (1 - p), std=lambda n, p: np.sqrt(n * p * (1 - p)),
     skewness=lambda n, p: ((1 - p) - p) / np.sqrt(n * p * (1 - p)),
     kurtosis=lambda n, p: (1 - 6 * p * (1 - p)) / (n * p * (1 - p)),
     entropy=lambda n, p: _per_set(lambda n_, p_: discrete_entropy(0, n_ + 1, _fn("binomial", "logpdf"), n_, p_), n, p))

# -------------------------------------------------------------------------------- exponential.py:5-38
_reg("exponential",
     mean=lambda lam: 1.0 / lam, mode=lambda lam: _full(0.0, lam), median=lambda lam: np.log(2.0) / lam,
     var=lambda lam: 1.0 / np.square(lam), std=lambda lam: 1.0 / lam, skewness=lambda lam: _full(2.0, lam),
     kurtosis=lambda lam: _full(6.0, lam), entropy=lambda lam: 1.0 - np.log(lam))

# ---------------------------------------------------------------------------------- lognormal.py:6-42
_reg("lognormal",
     mean=lambda mu, s: np.exp(mu + s**2 / 2), mode=lambda mu, s: np.exp(mu - s**2), median=lambda mu, s: _full(np.exp(mu), mu, s),
     var=lambda mu, s: (np.exp(s**2) - 1) * np.exp(2 * mu + s**2),
     std=lambda mu, s: np.sqrt((np.exp(s**2) - 1) * np.exp(2 * mu + s**2)),
     skewness=lambda mu, s: _full((np.exp(s**2) + 2) * np.sqrt(np.exp(s**2) - 1), mu, s),
     kurtosis=lambda mu, s: _full(np.exp(4 * s**2) + 2 * np.exp(3 * s**2) + 3 * np.exp(2 * s**2) - 6, mu, s),
     entropy=lambda mu, s: _full(np.log(np.sqrt(2 * np.pi) * s * np.exp(mu + 0.5)), mu, s))

# --------------------------------------------------------------------------------- halfnormal.py:5-35
_reg("halfnormal",
     mean=lambda s: s * np.sqrt(2 / np.pi), mode=np.zeros_like, median=lambda s: s * np.sqrt(2) * sp.erfinv(0.5),
     var=lambda s: s**2 * (1 - 2 / np.pi), std=lambda s: s * np.sqrt(1 - 2 / np.pi),
     skewness=lambda s: np.full_like(s, 0.9952717), kurtosis=lambda s: np.full_like(s, 0.8691773),
     entropy=lambda s: 0.5 * np.log(np.pi * s**2 / 2) + 0.5)

# ------------------------------------------------------------------------------------- cauchy.py:6-41
_reg("cauchy",
     mean=lambda a, b: _full(np.nan, a, b), mode=lambda a, b: _full(a, a, b), median=lambda a, b: _full(a, a, b),
     var=lambda a, b: _full(np.nan, a, b), std=lambda a, b: _full(np.nan, a, b), skewness=lambda a, b: _full(np.nan, a, b),
     kurtosis=lambda a, b: _full(np.nan, a, b), entropy=lambda a, b: np.log(4 * np.pi * _full(b, a, b)))

# --------------------------------------------------------------------------------- halfcauchy.py:6-35
_reg("halfcauchy",
     mean=lambda b: np.full_like(b, np.inf), mode=np.zeros_like, median=lambda b: _fn("halfcauchy", "ppf")(0.5, b),
     var=lambda b: np.full_like(b, np.inf), std=lambda b: np.full_like(b, np.inf), skewness=lambda b: np.full_like(b, np.nan),
     kurtosis=lambda b: np.full_like(b, np.nan), entropy=lambda b: np.log(2 * np.pi * b))

# ------------------------------------------------------------------------------------ laplace.py:6-43
_reg("laplace",
     mean=lambda mu, b: _full(mu, mu, b), mode=lambda mu, b: _full(mu, mu, b), median=lambda mu, b: _full(mu, mu, b),
     var=lambda mu, b: 2.0 * np.square(_full(b, mu, b)), std=lambda mu, b: np.sqrt(2) * _full(b, mu, b),
     skewness=lambda mu, b: _full(0.0, mu, b), kurtosis=lambda mu, b: _full(3.0, mu, b),
     entropy=lambda mu, b: np.log(2.0 * _full(b, mu, b)) + 1.0)

# ----------------------------------------------------------------------------------- logistic.py:6-43
_reg("logistic",
     mean=lambda mu, s: _full(mu, mu, s), mode=lambda mu, s: _full(mu, mu, s), median=lambda mu, s: _full(mu, mu, s),
     var=lambda mu, s: _full(s**2 * np.pi**2 / 3, mu, s), std=lambda mu, s: _full(s * np.pi / np.sqrt(3), mu, s),
     skewness=lambda mu, s: _full(0.0, mu, s), kurtosis=lambda mu, s: _full(1.2, mu, s),
     entropy=lambda mu, s: _full(np.log(s) + 2, mu, s))

# ------------------------------------------------------------------------------------- gumbel.py:6-41
_reg("gumbel",
     mean=lambda mu, b: mu + b * EULER, mode=lambda mu, b: _full(mu, mu, b), median=lambda mu, b: mu - b * -0.3665129,
     var=lambda mu, b: _full(1.64493407 * b**2, mu, b), std=lambda mu, b: _full(b * 1.2825498, mu, b),
     skewness=lambda mu, b: _full(1.1395471, mu, b), kurtosis=lambda mu, b: _full(2.4, mu, b),
     entropy=lambda mu, b: _full(np.log(b) + 1 + EULER, mu, b))


# ------------------------------------------------------------------------------------ weibull.py:8-44
def _wb_mean(a, b):
    return b * sp.gamma(1 + 1 / a)


def _wb_var(a, b):
    return b**2 * sp.gamma(1 + 2 / a) - _wb_mean(a, b) ** 2


def _wb_skew(a, b):
    sigma = np.sqrt(_wb_var(a, b))
    m_s = _wb_mean(a, b) / sigma
    return sp.gamma(1 + 3 / a) * (b / sigma) ** 3 - 3 * m_s - m_s**3


def _wb_kurt(a, b):
    sigma = np.sqrt(_wb_var(a, b))
    m_s = _wb_mean(a, b) / sigma
    return (b / sigma) ** 4 * sp.gamma(1 + 4 / a) - 4 * _wb_skew(a, b) * m_s - 6 * m_s**2 - m_s**4 - 3


_reg("weibull",
     mean=_wb_mean, mode=lambda a, b: np.where(a > 1, b * ((a - 1) / a) ** (1 / a), 0.0),
     median=lambda a, b: b * np.log(2) ** (1 / a), var=_wb_var, std=lambda a, b: np.sqrt(_wb_var(a, b)),
     skewness=_wb_skew, kurtosis=_wb_kurt, entropy=lambda a, b: EULER * (1 - 1 / a) + np.log(b / a) + 1)

# ------------------------------------------------------------------------------------- pareto.py:5-44
_reg("pareto",
     mean=lambda a, m: np.where(a > 1, a * m / (a - 1), np.inf), mode=lambda a, m: _full(m, a, m),
     median=lambda a, m: m * 2 ** (1 / a),
     var=lambda a, m: np.where(a > 2, m**2 * a / ((a - 1) ** 2 * (a - 2)), np.inf),
     std=lambda a, m: np.sqrt(np.where(a > 2, m**2 * a / ((a - 1) ** 2 * (a - 2)), np.inf)),
     skewness=lambda a, m: np.where(_full(a, a, m) > 3, 2 * (1 + _full(a, a, m)) / (_full(a, a, m) - 3) * np.sqrt(1 - 2 / _full(a, a, m)), np.nan),
     kurtosis=lambda a, m: (lambda A: np.where(A > 4, 6 * (A**3 + A**2 - 6 * A - 2) / (A * (A - 3) * (A - 4)), np.nan))(_full(a, a, m)),
     entropy=lambda a, m: np.log(m / a) + (1 + 1 / a))

# ------------------------------------------------------------------------------------ uniform.py:5-38
_reg("uniform",
     mean=lambda lo, up: (up + lo) / 2, mode=lambda lo, up: _full(np.nan, lo, up), median=lambda lo, up: (up + lo) / 2,
     var=lambda lo, up: (up - lo) ** 2 / 12, std=lambda lo, up: np.sqrt((up - lo) ** 2 / 12),
     skewness=lambda lo, up: _full(0.0, lo, up), kurtosis=lambda lo, up: _full(-6 / 5, lo, up),
     entropy=lambda lo, up: np.log(up - lo))


# --------------------------------------------------------------------------------- triangular.py:5-41
def _tri_q(lo, c, up):
    return lo**2 + up**2 + c**2 - lo * c - c * up - lo * up


_reg("triangular",
     mean=lambda lo, c, up: (lo + c + up) / 3, mode=lambda lo, c, up: _full(c, lo, c, up),
     median=lambda lo, c, up: np.where(c >= (lo + up) / 2, lo + np.sqrt((up - lo) * (c - lo) / 2), up - np.sqrt((up - lo) * (up - c) / 2)),
     var=lambda lo, c, up: _tri_q(lo, c, up) / 18, std=lambda lo, c, up: np.sqrt(_tri_q(lo, c, up) / 18),
     skewness=lambda lo, c, up: (np.sqrt(2) * (lo + up - 2 * c) * (2 * lo - up - c) * (lo - 2 * up + c)) / (5 * np.power(_tri_q(lo, c, up), 1.5)),
     kurtosis=lambda lo, c, up: _full(-3 / 5, lo, c, up),
     entropy=lambda lo, c, up: 0.5 + np.log((_full(up, lo, c, up) - _full(lo, lo, c, up)) / 2))


# -------------------------------------------------------------------------------- kumaraswamy.py:6-50
def _ku_m(k, a, b):
    return b * np.exp(sp.gammaln(1 + k / a) + sp.gammaln(b) - sp.gammaln(1 + k / a + b))


def _ku_var(a, b):
    return _ku_m(2, a, b) - _ku_m(1, a, b) ** 2


def _ku_mode(a, b):
    a, b = np.broadcast_arrays(a, b)
    return np.where((a > 1) | (b > 1), ((a - 1) / (a * b - 1)) ** (1 / a), 0.5)


def _ku_kurt(a, b):
    m1, v = _ku_m(1, a, b), _ku_var(a, b)
    m2, m3, m4 = _ku_m(2, a, b), _ku_m(3, a, b), _ku_m(4, a, b)
    return (m4 + m1 * (-4 * m3 + m1 * (6 * m2 - 3 * m1**2))) / v**2 - 3


_reg("kumaraswamy",
     mean=lambda a, b: _ku_m(1, a, b), mode=_ku_mode, median=lambda a, b: _fn("kumaraswamy", "ppf")(0.5, a, b),
     var=_ku_var, std=lambda a, b: np.sqrt(_ku_var(a, b)),
     skewness=lambda a, b: (_ku_m(3, a, b) - 3 * _ku_m(1, a, b) * _ku_var(a, b) - _ku_m(1, a, b) ** 3) / _ku_var(a, b) ** 1.5,
     kurtosis=_ku_kurt,
     entropy=lambda a, b: (1 - 1 / b) + (1 - 1 / a) * (sp.psi(b + 1) + EULER) - np.log(a) - np.log(b))

# -------------------------------------------------------------------------------------- moyal.py:5-40
_reg("moyal",
     mean=lambda mu, s: mu + s * (EULER + np.log(2)), mode=lambda mu, s: _full(mu, mu, s),
     median=lambda mu, s: _full(mu + s * 0.7875976, mu, s), var=lambda mu, s: _full(np.square(s) * (np.pi**2) / 2, mu, s),
     std=lambda mu, s: np.sqrt(_full(np.square(s) * (np.pi**2) / 2, mu, s)), skewness=lambda mu, s: _full(1.5351416, mu, s),
     kurtosis=lambda mu, s: _full(4.0, mu, s), entropy=lambda mu, s: np.log(s) + 2.0541199)

# -------------------------------------------------------------------------- asymmetriclaplace.py:5-40
_reg("asymmetriclaplace",
     mean=lambda mu, b, k: (1 / k - k) * b + mu, mode=lambda mu, b, k: _full(mu, mu, b, k),
     median=lambda mu, b, k: np.where(k > 1, mu + k * b * np.log((1 + k**2) / (2 * k**2)), mu - np.log((1 + k**2) / 2) / (k / b)),
     var=lambda mu, b, k: ((1 / k) ** 2 + k**2) * b**2, std=lambda mu, b, k: np.sqrt(((1 / k) ** 2 + k**2) * b**2),
     skewness=lambda mu, b, k: 2.0 * (1 - np.power(k, 6)) / np.power(1 + np.power(k, 4), 1.5),
     kurtosis=lambda mu, b, k: 6.0 * (1 + np.power(k, 8)) / np.power(1 + np.power(k, 4), 2),
     entropy=lambda mu, b, k: 1 + np.log(k + 1 / k) + np.log(b))


# -------------------------------------------------------------------------------- loglogistic.py:5-63
def _ll_parts(a, be):
    b = np.pi / be
    m = [None] + [k * b / np.sin(k * b) for k in (1, 2, 3, 4)]
    mu = a * m[1]
    sigma2 = a**2 * (m[2] - m[1] ** 2)
    return b, m, mu, sigma2, np.sqrt(sigma2)


def _ll_skew_raw(a, be):
    b, m, mu, s2, s = _ll_parts(a, be)
    return (a**3 * m[3] - 3 * mu * s2 - mu**3) / s**3


def _ll_kurt(a, be):
    b, m, mu, s2, s = _ll_parts(a, be)
    m_s = mu / s
    r = (a / s) ** 4 * m[4] - 4 * _ll_skew_raw(a, be) * m_s - 6 * m_s**2 - m_s**4 - 3
    return np.where(be > 4, r, np.nan)


def _ll_var(a, be):
    b = np.pi / be
    return np.where(be > 2, a**2 * (2 * b / np.sin(2 * b) - b**2 / np.sin(b) ** 2), np.nan)


_reg("loglogistic",
     mean=lambda a, be: np.where(be > 1, a * (np.pi / be) / np.sin(np.pi / be), np.nan),
     mode=lambda a, be: np.where(be > 1, a * ((be - 1) / (be + 1)) ** (1 / be), np.zeros_like(a * be)),
     median=lambda a, be: _full(a, a, be), var=_ll_var, std=lambda a, be: np.sqrt(_ll_var(a, be)),
     skewness=lambda a, be: np.where(be > 3, _ll_skew_raw(a, be), np.nan), kurtosis=_ll_kurt,
     entropy=lambda a, be: np.log(a) - np.log(be) + 2)


# ------------------------------------------------------------------------------- logitnormal.py:12-92
def _ghq(mu, s, order=1, mean_val=None, n_points=70):
    gh_x, gh_w = np.polynomial.hermite.hermgauss(n_points)
    shape = np.broadcast_shapes(np.shape(mu), np.shape(s))
    gx = gh_x.reshape((-1,) + (1,) * len(shape))
    gw = gh_w.reshape((-1,) + (1,) * len(shape))
    x = sp.expit(np.sqrt(2.0) * s * gx + mu)
    integrand = (x - mean_val) ** order if mean_val is not None else x**order
    return np.sum(gw * integrand, axis=0) / np.sqrt(np.pi)


def _ln_var(mu, s):
    return _ghq(mu, s, 2, _ghq(mu, s, 1))


def _ln_entropy(mu, s):
    gh_x, gh_w = np.polynomial.hermite.hermgauss(70)
    shape = np.broadcast_shapes(np.shape(mu), np.shape(s))
    gx = gh_x.reshape((-1,) + (1,) * len(shape))
    gw = gh_w.reshape((-1,) + (1,) * len(shape))
    x = sp.expit(np.sqrt(2.0) * s * gx + mu)
    return np.sum(gw * -_fn("logitnormal", "logpdf")(x, mu, s), axis=0) / np.sqrt(np.pi)


_reg("logitnormal",
     mean=lambda mu, s: _ghq(mu, s, 1), mode=lambda mu, s: sp.expit(mu), median=lambda mu, s: _full(sp.expit(mu), mu, s),
     var=_ln_var, std=lambda mu, s: np.sqrt(_ln_var(mu, s)),
     skewness=lambda mu, s: _ghq(mu, s, 3, _ghq(mu, s, 1)) / (np.sqrt(_ln_var(mu, s)) ** 3),
     kurtosis=lambda mu, s: _ghq(mu, s, 4, _ghq(mu, s, 1)) / (_ln_var(mu, s) ** 2) - 3, entropy=_ln_entropy)


# ------------------------------------------------------------------------------ inversegamma.py:7-45
def _ig_var(a, b):
    return np.where(a > 2, b**2 / ((a - 1) ** 2 * (a - 2)), np.inf)


def _ig_entropy(a, b):
    regular = a - (a + 1.0) * sp.digamma(a) + sp.gammaln(a) + np.log(b)
    asym = ((1 - 3 * np.log(a) + np.log(2) + np.log(np.pi)) / 2 + 2 / 3 * a**-1.0 + a**-2.0 / 12 - a**-3.0 / 90
            - a**-4.0 / 120 + np.log(b))
    return np.where(a >= 200, asym, regular)


_reg("inversegamma",
     mean=lambda a, b: np.where(a > 1, b / (a - 1), np.inf), mode=lambda a, b: b / (a + 1),
     median=lambda a, b: _fn("inversegamma", "ppf")(0.5, a, b), var=_ig_var, std=lambda a, b: np.sqrt(_ig_var(a, b)),
     skewness=lambda a, b: np.where(a > 3, 4 * np.sqrt(a - 2) / (a - 3), np.nan) + 0 * b,
     kurtosis=lambda a, b: np.where(a > 4, 6 * (5 * a - 11) / ((a - 3) * (a - 4)), np.nan) + 0 * b, entropy=_ig_entropy)

# -------------------------------------------------------------------------------- chisquared.py:6-37
_reg("chisquared",
     mean=lambda nu: nu * 1.0, mode=lambda nu: np.maximum(nu - 2, 0), median=lambda nu: _fn("chisquared", "ppf")(0.5, nu),
     var=lambda nu: nu * 2, std=lambda nu: np.sqrt(nu * 2), skewness=lambda nu: np.sqrt(8 / nu), kurtosis=lambda nu: 12 / nu,
     entropy=lambda nu: nu / 2 + np.log(2) + sp.gammaln(nu / 2) + (1 - nu / 2) * sp.digamma(nu / 2))


# ------------------------------------------------------------------- scaledinversechisquared.py:6-40
def _sic_var(nu, t2):
    return np.where(nu > 4, 2 * nu**2 * t2**2 / ((nu - 2) ** 2 * (nu - 4)), np.inf)


_reg("scaledinversechisquared",
     mean=lambda nu, t2: np.where(nu > 2, nu * t2 / (nu - 2), np.inf), mode=lambda nu, t2: nu * t2 / (nu + 2),
     median=lambda nu, t2: _fn("scaledinversechisquared", "ppf")(0.5, nu, t2), var=_sic_var,
     std=lambda nu, t2: np.sqrt(_sic_var(nu, t2)),
     skewness=lambda nu, t2: np.where(nu > 6, 4 / (nu - 6) * np.sqrt(2 * (nu - 4)), np.nan) + 0 * t2,
     kurtosis=lambda nu, t2: np.where(nu > 8, (12 * (5 * nu - 22)) / ((nu - 6) * (nu - 8)), np.nan) + 0 * t2,
     entropy=lambda nu, t2: nu / 2 + np.log(nu / 2 * t2) + sp.gammaln(nu / 2) - (1 + nu / 2) * sp.digamma(nu / 2))


# ------------------------------------------------------------------------------ halfstudentt.py:11-73
def _hst_g(nu):
    g0, g1 = np.exp(sp.gammaln((nu + 1) / 2)), np.exp(sp.gammaln(nu / 2))
    return g0, g1, np.isinf(g0) | np.isinf(g1)


def _hst_mean(nu, s):
    g0, g1, bad = _hst_g(nu)
    fin = 2 * s * np.sqrt(nu / np.pi) * (g0 / (g1 * (nu - 1)))
    return np.where(nu > 1, np.where(bad, s * np.sqrt(2 / np.pi), fin), np.inf)


def _hst_var(nu, s):
    g0, g1, bad = _hst_g(nu)
    fin = s**2 * ((nu / (nu - 2)) - ((4 * nu) / (np.pi * (nu - 1) ** 2)) * (g0 / g1) ** 2)
    return np.where(nu > 2, np.where(bad, s**2 * (1 - 2.0 / np.pi), fin), np.inf)


def _hst_std(nu, s):
    v = _hst_var(nu, s)
    return np.where(np.isinf(v) | np.isnan(v), v, np.sqrt(v))


def _hst_entropy(nu, s):
    t = (np.log(s) + 0.5 * (nu + 1) * (sp.psi(0.5 * (nu + 1)) - sp.psi(0.5 * nu)) + np.log(np.sqrt(nu)) + R.betaln(0.5 * nu, 0.5)
         - np.log(2))
    return np.where(nu > 1e5, 0.5 * np.log(np.pi * s**2 / 2) + 0.5, t)


_reg("halfstudentt",
     mean=_hst_mean, mode=lambda nu, s: np.zeros_like(_full(s, nu, s)), median=lambda nu, s: _fn("halfstudentt", "ppf")(0.5, nu, s),
     var=_hst_var, std=_hst_std, skewness=lambda nu, s: _full(np.nan, nu, s), kurtosis=lambda nu, s: _full(np.nan, nu, s),
     entropy=_hst_entropy)


# --------------------------------------------------------------------------------- betascaled.py:9-68
def _bs_mode(a, b, lo, up):
    a, b, lo, up = np.broadcast_arrays(a, b, lo, up)
    r = np.full(a.shape, np.nan)
    r = np.where((a <= 1) & (b > 1), lo, r)
    r = np.where((b <= 1) & (a > 1), up, r)
    return np.where((a > 1) & (b > 1), lo + (a - 1) / (a + b - 2) * (up - lo), r)


def _bs_var(a, b, lo, up):
    return (a * b) / ((a + b) ** 2 * (a + b + 1)) * (up - lo) ** 2


_reg("betascaled",
     mean=lambda a, b, lo, up: (a * up + b * lo) / (a + b), mode=_bs_mode,
     median=lambda a, b, lo, up: _fn("betascaled", "ppf")(0.5, a, b, lo, up), var=_bs_var,
     std=lambda a, b, lo, up: np.sqrt(_bs_var(a, b, lo, up)),
     skewness=lambda a, b, lo, up: MOMENTS["beta"]["skewness"](a, b) + 0 * (lo + up),
     kurtosis=lambda a, b, lo, up: _beta_kurt(a, b) + 0 * (lo + up),
     entropy=lambda a, b, lo, up: MOMENTS["beta"]["entropy"](a, b) + np.log(up - lo))


# ---------------------------------------------------------------------------------- bernoulli.py:8-38
def _xlogx(x):
    with np.errstate(all="ignore"):
        return np.where(x == 0, 0.0, x * np.log(x))


_reg("bernoulli",
     mean=lambda p: p * 1.0, mode=lambda p: np.where(p <= 0.5, 0.0, 1.0), median=lambda p: np.where(p <= 0.5, 0.0, 1.0),
     var=lambda p: p * (1 - p), std=lambda p: np.sqrt(p * (1 - p)), skewness=lambda p: ((1 - p) - p) / np.sqrt(p * (1 - p)),
     kurtosis=lambda p: (1 - 6 * p * (1 - p)) / (p * (1 - p)), entropy=lambda p: -_xlogx(p) - _xlogx(1 - p))

# ---------------------------------------------------------------------------------- geometric.py:6-35
_reg("geometric",
     mean=lambda p: 1 / p, mode=np.ones_like, median=lambda p: _fn("geometric", "ppf")(0.5, p), var=lambda p: (1 - p) / p**2,
     std=lambda p: np.sqrt((1 - p) / p**2), skewness=lambda p: (2 - p) / np.sqrt(1 - p), kurtosis=lambda p: 6 + (p**2) / (1 - p),
     entropy=lambda p: (-(1 - p) * np.log(1 - p) - p * np.log(p)) / p)

# ---------------------------------------------------------------------------- discreteuniform.py:6-38
_reg("discreteuniform",
     mean=lambda lo, up: (up + lo) / 2, mode=lambda lo, up: _full(np.nan, lo, up), median=lambda lo, up: np.floor((up + lo) / 2),
     var=lambda lo, up: ((up - lo + 1) ** 2 - 1) / 12, std=lambda lo, up: np.sqrt(((up - lo + 1) ** 2 - 1) / 12),
     skewness=lambda lo, up: np.zeros_like(lo) + 0 * up,
     kurtosis=lambda lo, up: -(6 * ((up - lo + 1) ** 2 + 1)) / (5 * ((up - lo + 1) ** 2 - 1)),
     entropy=lambda lo, up: _per_set(lambda a, b: discrete_entropy(a, b + 1, _fn("discreteuniform", "logpdf"), a, b), lo, up))


# --------------------------------------------------------------------------- discreteweibull.py:14-46
def _dw(order):
    return lambda q, b: _per_set(lambda q_, b_: discrete_moment(_fn("discreteweibull", "ppf"), _fn("discreteweibull", "pdf"), q_, b_, order=order), q, b)


def _dw_entropy(q, b):
    def one(q_, b_):
        upper = min(float(_fn("discreteweibull", "ppf")(np.float64(0.9999), q_, b_)), 1e4)
        return discrete_entropy(0, upper, _fn("discreteweibull", "logpdf"), q_, b_)
    return _per_set(one, q, b)


_reg("discreteweibull",
     mean=_dw(1), mode=lambda q, b: np.floor(-np.log(-np.log(q)) / np.log(b)), median=lambda q, b: _fn("discreteweibull", "ppf")(0.5, q, b),
     var=_dw(2), std=lambda q, b: np.sqrt(_dw(2)(q, b)), skewness=lambda q, b: _dw(3)(q, b) / (_dw(2)(q, b) ** 1.5),
     kurtosis=lambda q, b: _dw(4)(q, b) / (_dw(2)(q, b) ** 2) - 3, entropy=_dw_entropy)


# --------------------------------------------------------------------------- negativebinomial.py:7-39
def _nb_entropy(n, p):
    def one(n_, p_):
        ppf = _fn("negativebinomial", "ppf")
        return discrete_entropy(float(ppf(np.float64(0.0001), n_, p_)), float(ppf(np.float64(0.9999), n_, p_)),
                                _fn("negativebinomial", "logpdf"), n_, p_)
    return _per_set(one, n, p)


_reg("negativebinomial",
     mean=lambda n, p: n * (1 - p) / p, mode=lambda n, p: np.where(n < 1, 0.0, np.floor((n - 1) * (1 - p) / p)),
     median=lambda n, p: _fn("negativebinomial", "ppf")(0.5, n, p), var=lambda n, p: n * (1 - p) / (p * p),
     std=lambda n, p: np.sqrt(n * (1 - p) / (p * p)), skewness=lambda n, p: (2 - p) / np.sqrt(n * (1 - p)),
     kurtosis=lambda n, p: 6.0 / n + (p * p) / (n * (1 - p)), entropy=_nb_entropy)


# ------------------------------------------------------------------------------ betabinomial.py:10-80
def _bb_mode(n, a, b):
    n, a, b = np.broadcast_arrays(n, a, b)
    r = np.full(a.shape, np.nan)
    r = np.where((a == 1) & (b > 1), 0.0, r)
    r = np.where((a > 1) & (b == 1), n, r)
    std_mode = np.floor((n + 1) * ((a - 1) / (a + b - 2)))
    return np.where((a > 1) & (b > 1), np.clip(std_mode, 0, n), r)


def _bb_var(n, a, b):
    return n * a * b * (a + b + n) / ((a + b) ** 2 * (a + b + 1))


def _bb_kurt(n, a, b):
    s, pr = a + b, a * b
    left = ((s) ** 2 * (1 + s)) / ((n * pr) * (s + 2) * (s + 3) * (s + n))
    right = s * (s - 1 + 6 * n) + 3 * pr * (n - 2) + 6 * n**2
    right = right - (3 * pr * n * (6 - n)) / s
    right = right - (18 * pr * n**2) / (s) ** 2
    return (left * right) - 3


_reg("betabinomial",
     mean=lambda n, a, b: n * a / (a + b), mode=_bb_mode, median=lambda n, a, b: _fn("betabinomial", "ppf")(0.5, n, a, b),
     var=_bb_var, std=lambda n, a, b: np.sqrt(_bb_var(n, a, b)),
     skewness=lambda n, a, b: (a + b + 2 * n) * (b - a) / (a + b + 2) * np.sqrt((1 + a + b) / (n * a * b * (a + b + n))),
     kurtosis=_bb_kurt,
     entropy=lambda n, a, b: _per_set(lambda n_, a_, b_: discrete_entropy(0, n_ + 1, _fn("betabinomial", "logpdf"), n_, a_, b_), n, a, b))


# ---------------------------------------------------------------------------- hypergeometric.py:8-60
def _hg_var(N, k, n):
    return n * k * (N - k) * (N - n) / (N * N * (N - 1))


def _hg_kurt(N, k, n):
    m = N - k
    num = N * N * (N - 1) * (N * (N + 1) - 6 * n * (N - n) - 6 * k * m) + 6 * k * n * (N - n) * m * (5 * N - 6)
    return num / (k * n * (N - n) * m * (N - 2) * (N - 3))


_reg("hypergeometric",
     mean=lambda N, k, n: n * k / N, mode=lambda N, k, n: np.floor((n + 1) * (k + 1) / (N + 2)),
     median=lambda N, k, n: _fn("hypergeometric", "ppf")(0.5, N, k, n), var=_hg_var, std=lambda N, k, n: np.sqrt(_hg_var(N, k, n)),
     skewness=lambda N, k, n: ((N - 2 * k) * np.sqrt(N - 1) * (N - 2 * n)) / (np.sqrt(n * k * (N - k) * (N - n)) * (N - 2)),
     kurtosis=_hg_kurt,
     entropy=lambda N, k, n: _per_set(lambda N_, k_, n_: discrete_entropy(max(0, n_ + k_ - N_), min(k_, n_) + 1, _fn("hypergeometric", "logpdf"),
                                                                         N_, k_, n_), N, k, n))


# --------------------------------------------------------------------------------- helper.py:425-446
def _zi_mode(name, base_mode, *params):
    lp = _fn(name, "logpdf")
    return np.where(lp(0.0, *params) >= lp(base_mode, *params), 0.0, base_mode)


def _central_from_raw(ex1, ex2, ex3, ex4=None):
    mu2 = ex2 - np.power(ex1, 2)
    if ex4 is None:
        return (ex3 - 3 * ex1 * ex2 + 2 * np.power(ex1, 3)) / np.power(mu2, 1.5)
    mu4 = ex4 - 4 * ex1 * ex3 + 6 * np.power(ex1, 2) * ex2 - 3 * np.power(ex1, 4)
    return mu4 / np.power(mu2, 2) - 3


# --------------------------------------------------------------------------------- zi_poisson.py:9-59
def _zip_entropy(psi, mu):
    def one(psi_, mu_):
        lp = _fn("zi_poisson", "logpdf")
        return discrete_entropy(0, 25, lp, psi_, mu_) if mu_ < 10 else discrete_entropy(0, int(mu_ * 3), lp, psi_, mu_)
    return _per_set(one, psi, mu)


_reg("zi_poisson",
     mean=lambda psi, mu: psi * mu, mode=lambda psi, mu: _zi_mode("zi_poisson", np.floor(mu), psi, mu),
     median=lambda psi, mu: _fn("zi_poisson", "ppf")(0.5, psi, mu), var=lambda psi, mu: psi * (mu + (1 - psi) * np.power(mu, 2)),
     std=lambda psi, mu: np.sqrt(psi * (mu + (1 - psi) * np.power(mu, 2))),
     skewness=lambda psi, mu: _central_from_raw(psi * mu, psi * (mu + mu**2), psi * (mu + 3 * mu**2 + mu**3)),
     kurtosis=lambda psi, mu: _central_from_raw(psi * mu, psi * (mu + mu**2), psi * (mu + 3 * mu**2 + mu**3),
                                                psi * (mu + 7 * mu**2 + 6 * mu**3 + mu**4)),
     entropy=_zip_entropy)


# -------------------------------------------------------------------------------- zi_binomial.py:9-72
def _zib_var(psi, n, p):
    return psi * (n * p * (1 - p) + (1 - psi) * np.power(n * p, 2))


def _zib_skew(psi, n, p):
    q = 1 - p
    b1 = n * p
    b2 = n * p * q + np.power(b1, 2)
    b3 = n * p * q * (1 - 2 * p) + 3 * np.power(n * p, 2) * q + np.power(b1, 3)
    return _central_from_raw(psi * b1, psi * b2, psi * b3)


def _fact_raw(f1, f2, f3, f4):
    return f1, f2 + f1, f3 + 3 * f2 + f1, f4 + 6 * f3 + 7 * f2 + f1


def _zib_kurt(psi, n, p):
    b = _fact_raw(n * p, n * (n - 1) * p**2, n * (n - 1) * (n - 2) * p**3, n * (n - 1) * (n - 2) * (n - 3) * p**4)
    return _central_from_raw(*[psi * v for v in b])


_reg("zi_binomial",
     mean=lambda psi, n, p: psi * n * p, mode=lambda psi, n, p: _zi_mode("zi_binomial", np.floor((n + 1) * p), psi, n, p),
     median=lambda psi, n, p: _fn("zi_binomial", "ppf")(0.5, psi, n, p), var=_zib_var, std=lambda psi, n, p: np.sqrt(_zib_var(psi, n, p)),
     skewness=_zib_skew, kurtosis=_zib_kurt,
     entropy=lambda psi, n, p: _per_set(lambda a, b, c: discrete_entropy(0, b + 1, _fn("zi_binomial", "logpdf"), a, b, c), psi, n, p))


# ------------------------------------------------------------------------ zi_negativebinomial.py:9-79
def _zinb_var(psi, n, p):
    return psi * (n * (1 - p) / np.power(p, 2) + (1 - psi) * np.power(n * (1 - p) / p, 2))


def _zinb_raw(n, p):
    q = 1 - p
    return _fact_raw(n * q / p, n * (n + 1) * q**2 / p**2, n * (n + 1) * (n + 2) * q**3 / p**3,
                     n * (n + 1) * (n + 2) * (n + 3) * q**4 / p**4)


def _zinb_entropy(psi, n, p):
    def one(a, b, c):
        ppf = _fn("zi_negativebinomial", "ppf")
        return discrete_entropy(float(ppf(np.float64(0.0001), a, b, c)), float(ppf(np.float64(0.9999), a, b, c)),
                                _fn("zi_negativebinomial", "logpdf"), a, b, c)
    return _per_set(one, psi, n, p)


_reg("zi_negativebinomial",
     mean=lambda psi, n, p: psi * n * (1 - p) / p,
     mode=lambda psi, n, p: _zi_mode("zi_negativebinomial", np.where(n < 1, 0.0, np.floor((n - 1) * (1 - p) / p)), psi, n, p),
     median=lambda psi, n, p: _fn("zi_negativebinomial", "ppf")(0.5, psi, n, p), var=_zinb_var,
     std=lambda psi, n, p: np.sqrt(_zinb_var(psi, n, p)),
     skewness=lambda psi, n, p: _central_from_raw(*[psi * v for v in _zinb_raw(n, p)[:3]]),
     kurtosis=lambda psi, n, p: _central_from_raw(*[psi * v for v in _zinb_raw(n, p)]), entropy=_zinb_entropy)


# -------------------------------------------------------------------------------------- wald.py:5-40
def _wald_entropy(mu, lam):
    x = 2 * lam / mu
    gamma_term = -sp.gammaincc(1e-8, x) * sp.gamma(1e-8)
    return 0.5 * np.log((2 * np.pi * np.e * mu**3) / lam) + 1.5 * np.exp(x) * gamma_term


_reg("wald",
     mean=lambda mu, lam: _full(mu, mu, lam),
     mode=lambda mu, lam: mu * (np.sqrt(1 + (3 * mu / (2 * lam)) ** 2) - 3 * mu / (2 * lam)),
     median=lambda mu, lam: _fn("wald", "ppf")(0.5, mu, lam), var=lambda mu, lam: mu**3 / lam, std=lambda mu, lam: np.sqrt(mu**3 / lam),
     skewness=lambda mu, lam: 3.0 * np.sqrt(mu / lam), kurtosis=lambda mu, lam: 15.0 * mu / lam, entropy=_wald_entropy)


# -------------------------------------------------------------------------------- exgaussian.py:8-50
def _exg_mode(mu, s, nu):
    tau = 1 / nu
    return mu - np.sign(tau) * np.sqrt(2) * s * sp.erfcinv(np.abs(tau) / s * np.sqrt(2 / np.pi)) + s**2 / tau


def _exg_entropy(mu, s, nu):
    def one(a, b, c):
        ppf = _fn("exgaussian", "ppf")
        return continuous_entropy(float(ppf(np.float64(0.0001), a, b, c)), float(ppf(np.float64(0.9999), a, b, c)),
                                  _fn("exgaussian", "logpdf"), a, b, c)
    return _per_set(one, mu, s, nu)


_reg("exgaussian",
     mean=lambda mu, s, nu: mu + nu + 0 * s, mode=_exg_mode, median=lambda mu, s, nu: _fn("exgaussian", "ppf")(0.5, mu, s, nu),
     var=lambda mu, s, nu: np.square(s) + np.square(nu) + 0 * mu, std=lambda mu, s, nu: np.sqrt(np.square(s) + np.square(nu)) + 0 * mu,
     skewness=lambda mu, s, nu: 2 * np.power(nu / s, 3) * np.power(1.0 + np.square(nu / s), -1.5) + 0 * mu,
     kurtosis=lambda mu, s, nu: 6.0 * np.square(np.square(nu / s)) * np.power(1.0 + np.square(nu / s), -2) + 0 * mu,
     entropy=_exg_entropy)


from . import restatement_solver as S  # noqa: E402


def continuous_mode(lower, upper, logpdf, *params, n_points=200):
    """helper.py:375-422, one parameter set: the grid point with the largest logpdf (first maximum)."""
    x = np.linspace(lower, upper, n_points)
    with np.errstate(all="ignore"):
        return float(x[int(np.argmax(logpdf(x, *params)))])


# --------------------------------------------------------------------------- truncatednormal.py:7-165
def _log_phi(z):
    return -0.5 * z**2 - np.log(np.sqrt(2 * np.pi))


def _tn_terms(mu, s, lo, up):
    al, be = S._alpha_beta(mu, s, lo, up)
    lz = S._log_Z(al, be)
    return al, be, lz, np.exp(_log_phi(al) - lz), np.exp(_log_phi(be) - lz)


def _tn_var(mu, s, lo, up):
    al, be, lz, ta, tb = _tn_terms(mu, s, lo, up)
    return s**2 * (1 + (al * ta - be * tb) - np.square(ta - tb))


def _tn_raw(mu, s, lo, up):
    al, be, lz, pa, pb = _tn_terms(mu, s, lo, up)
    m1 = pa - pb
    m2 = 1 + al * pa - be * pb
    m3 = 2 * m1 + al**2 * pa - be**2 * pb
    m4 = 3 * m2 + al**3 * pa - be**3 * pb
    return m1, m2, m3, m4


def _tn_skew(mu, s, lo, up):
    m1, m2, m3, _ = _tn_raw(mu, s, lo, up)
    return (m3 - 3 * m1 * m2 + 2 * m1**3) / np.power(m2 - m1**2, 1.5)


def _tn_kurt(mu, s, lo, up):
    m1, m2, m3, m4 = _tn_raw(mu, s, lo, up)
    return (m4 - 4 * m1 * m3 + 6 * m1**2 * m2 - 3 * m1**4) / (m2 - m1**2) ** 2 - 3


def _tn_entropy(mu, s, lo, up):
    al, be, lz, ta, tb = _tn_terms(mu, s, lo, up)
    return np.log(np.sqrt(2 * np.pi * np.e) * s) + lz + 0.5 * (al * ta - be * tb)


_reg("truncatednormal",
     mean=lambda mu, s, lo, up: (lambda t: mu + s * (t[3] - t[4]))(_tn_terms(mu, s, lo, up)),
     mode=lambda mu, s, lo, up: np.clip(mu, lo, up) + 0 * s, median=lambda mu, s, lo, up: _fn("truncatednormal", "ppf")(0.5, mu, s, lo, up),
     var=_tn_var, std=lambda mu, s, lo, up: np.sqrt(_tn_var(mu, s, lo, up)), skewness=_tn_skew, kurtosis=_tn_kurt, entropy=_tn_entropy)


# ----------------------------------------------------------------------------- skew_studentt.py:9-101
def _sst_raw(n, a, b):
    bl = R.betaln
    log_coeff = 0.5 * n * np.log(a + b) - n * np.log(2) - bl(a, b)
    if n == 1:
        st = np.exp(bl(a + 0.5, b - 0.5)) - np.exp(bl(a - 0.5, b + 0.5))
    elif n == 2:
        st = np.exp(bl(a + 1, b - 1)) - 2 * np.exp(bl(a, b)) + np.exp(bl(a - 1, b + 1))
    elif n == 3:
        st = np.exp(bl(a + 1.5, b - 1.5)) - 3 * np.exp(bl(a + 0.5, b - 0.5)) + 3 * np.exp(bl(a - 0.5, b + 0.5)) - np.exp(bl(a - 1.5, b + 1.5))
    else:
        st = (np.exp(bl(a + 2, b - 2)) - 4 * np.exp(bl(a + 1, b - 1)) + 6 * np.exp(bl(a, b)) - 4 * np.exp(bl(a - 1, b + 1))
              + np.exp(bl(a - 2, b + 2)))
    return np.exp(log_coeff) * st


def _sst_var(a, b, mu, s):
    a, b, mu, s = np.broadcast_arrays(a, b, mu, s)
    return np.where((a > 1) & (b > 1), s**2 * (_sst_raw(2, a, b) - _sst_raw(1, a, b) ** 2), np.nan)


def _sst_std(a, b, mu, s):
    v = _sst_var(a, b, mu, s)
    return np.where(np.isinf(v) | np.isnan(v), v, np.sqrt(v))


def _sst_skew(a, b, mu, s):
    a, b, mu, s = np.broadcast_arrays(a, b, mu, s)
    e1, e2, e3 = _sst_raw(1, a, b), _sst_raw(2, a, b), _sst_raw(3, a, b)
    r = (e3 - 3 * e1 * e2 + 2 * e1**3) / np.sqrt(e2 - e1**2) ** 3
    return np.where((a > 1.5) & (b > 1.5), r, np.nan)


def _sst_kurt(a, b, mu, s):
    a, b, mu, s = np.broadcast_arrays(a, b, mu, s)
    e1, e2, e3, e4 = (_sst_raw(k, a, b) for k in (1, 2, 3, 4))
    r = (e4 - 4 * e1 * e3 + 6 * e1**2 * e2 - 3 * e1**4) / (e2 - e1**2) ** 2 - 3
    return np.where((a > 2) & (b > 2), r, np.nan)


def _sst_mode(a, b, mu, s):
    def one(a_, b_, mu_, s_):
        sd = s_ * np.sqrt((a_ + b_) / (2 * a_ * b_))
        return continuous_mode(mu_ - 5 * sd, mu_ + 5 * sd, _fn("skew_studentt", "logpdf"), a_, b_, mu_, s_)
    return _per_set(one, a, b, mu, s)


_reg("skew_studentt",
     mean=lambda a, b, mu, s: np.where((a > 0.5) & (b > 0.5), mu + s * _sst_raw(1, a, b), np.nan), mode=_sst_mode,
     median=lambda a, b, mu, s: _fn("skew_studentt", "ppf")(0.5, a, b, mu, s), var=_sst_var, std=_sst_std, skewness=_sst_skew,
     kurtosis=_sst_kurt,
     entropy=lambda a, b, mu, s: _per_set(lambda a_, b_, mu_, s_: continuous_entropy(mu_ - 50 * s_, mu_ + 50 * s_, _fn("skew_studentt", "logpdf"),
                                                                                      a_, b_, mu_, s_), a, b, mu, s))


# -------------------------------------------------------------------------------- skewnormal.py:8-59
def _sn_delta(alpha):
    return alpha / np.sqrt(1 + alpha**2)


def _sn_mode(mu, s, alpha):
    term1 = np.sqrt(2 / np.pi) * _sn_delta(alpha)
    term2 = (1 - np.pi / 4) * (term1**3) / (1 - (2 / np.pi) * _sn_delta(alpha) ** 2)
    term3 = 0.5 * np.sign(alpha) * np.exp(-2 * np.pi / np.abs(alpha))
    return mu + s * (term1 - term2 - term3)


def _sn_entropy(mu, s, alpha):
    h_normal = MOMENTS["normal"]["entropy"](mu, s)
    h_halfn = MOMENTS["halfnormal"]["entropy"](s)
    factor = 1 / np.sqrt(1 + alpha**2)
    return h_normal * factor + h_halfn * (1 - factor)


_reg("skewnormal",
     mean=lambda mu, s, al: mu + s * np.sqrt(2 / np.pi) * al / np.sqrt(1 + al**2), mode=_sn_mode,
     median=lambda mu, s, al: _fn("skewnormal", "ppf")(0.5, mu, s, al),
     var=lambda mu, s, al: s**2 * (1 - 2 * _sn_delta(al) ** 2 / np.pi) + 0 * mu,
     std=lambda mu, s, al: np.sqrt(s**2 * (1 - 2 * _sn_delta(al) ** 2 / np.pi)) + 0 * mu,
     skewness=lambda mu, s, al: (lambda mz: ((4 - np.pi) / 2) * (mz**3 / (1 - mz**2) ** (3 / 2)))(np.sqrt(2 / np.pi) * _sn_delta(al)) + 0 * (mu + s),
     kurtosis=lambda mu, s, al: 2 * (np.pi - 3) * ((_sn_delta(al) * np.sqrt(2 / np.pi)) ** 4 / (1 - 2 * _sn_delta(al) ** 2 / np.pi) ** 2) + 0 * (mu + s),
     entropy=_sn_entropy)


# ---------------------------------------------------------------------------------- vonmises.py:6-42
def _vm_var(mu, kappa):
    return 1 - sp.ive(1, kappa) / sp.ive(0, kappa) + 0 * mu


_reg("vonmises",
     mean=lambda mu, k: _full(mu, mu, k), mode=lambda mu, k: _full(mu, mu, k), median=lambda mu, k: _full(mu, mu, k),
     var=_vm_var, std=lambda mu, k: np.sqrt(_vm_var(mu, k)), skewness=lambda mu, k: _full(0.0, mu, k), kurtosis=lambda mu, k: _full(0.0, mu, k),
     entropy=lambda mu, k: np.log(2 * np.pi * sp.ive(0, k)) + k * _vm_var(mu, k))


# ------------------------------------------------------------------------------------- rice.py:13-60
def _rice_var(nu, s):
    lh = S._laguerre_half(nu**2 / (2 * s**2))
    return 2 * s**2 + nu**2 - (np.pi * s**2 / 2) * lh**2


def _rice_q(helper):
    return lambda nu, s: _per_set(lambda nu_, s_: helper(1e-10, nu_ + 10 * s_, _fn("rice", "logpdf"), nu_, s_), nu, s)


_reg("rice",
     mean=lambda nu, s: S._rice_mean(nu, s), mode=_rice_q(continuous_mode), median=lambda nu, s: _fn("rice", "ppf")(0.5, nu, s),
     var=_rice_var, std=lambda nu, s: np.sqrt(_rice_var(nu, s)), skewness=_rice_q(continuous_skewness), kurtosis=_rice_q(continuous_kurtosis),
     entropy=_rice_q(continuous_entropy))


# ------------------------------------------------------------------------------ polyagamma.py:15-111
def _pg_q(helper):
    def one(h_, z_):
        upper = float(S._pg_mean(h_, z_) + 10 * np.sqrt(S._pg_var(h_, z_)))
        return helper(1e-10, upper, _fn("polyagamma", "logpdf"), h_, z_)
    return lambda h, z: _per_set(one, h, z)


_reg("polyagamma",
     mean=lambda h, z: S._pg_mean(h, z), mode=_pg_q(continuous_mode), median=lambda h, z: _fn("polyagamma", "ppf")(0.5, h, z),
     var=lambda h, z: S._pg_var(h, z), std=lambda h, z: np.sqrt(S._pg_var(h, z)), skewness=_pg_q(continuous_skewness),
     kurtosis=_pg_q(continuous_kurtosis), entropy=_pg_q(continuous_entropy))
