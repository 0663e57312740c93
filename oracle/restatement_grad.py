"""CPU oracle for the gradient path -- TEST INFRASTRUCTURE ONLY.

The reference has no gradient code of its own: its functions are PyTensor graphs and
`pytensor.grad` differentiates them.  This file restates each `logpdf` graph (file:line cited per
function, paths relative to /root/reference/pytensor_distributions) on forward-mode dual numbers
over NumPy arrays, with PyTensor's derivative rules for the primitives it uses
(`log`, `log1p`, `exp`, `gammaln` -> `digamma`, `pow` -> [y x^(y-1), log(x) x^y], `abs` -> sgn,
`xlogy0(a, y)` -> [log y, a / y], `switch` -> the selected arm only).

PARITY UNPINNED BY THE REFERENCE: nothing in the reference's tests evaluates a gradient.  The
restatement is pinned (tests/test_oracle_grad.py) by (1) its value component reproducing the pinned
logpdf oracle (oracle/restatement.py) and (2) central finite differences of that pinned oracle.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from __future__ import annotations

import numpy as np
from scipy import special as sp

_F = np.float64
LOG_SQRT_2PI = 0.91893853320467274178
LOG_PI = 1.14472988584940017414
LOG_2 = 0.69314718055994530942


class Dual:
    """value + the partial derivatives that are structurally present: d maps a slot (0 = x, 1 + j = parameter j)
    to an array.  A slot that an expression does not depend on is absent -- never "0 * something" -- which is
    how reverse-mode differentiation of the graph (pytensor.grad) behaves: the gradient w.r.t. alpha only
    travels along paths that contain alpha, so an infinite factor on some other path cannot turn it into NaN."""

    __array_priority__ = 1000

    def __init__(self, v, d):
        self.v = np.asarray(v, dtype=_F)
        self.d = d

    @staticmethod
    def lift(o):
        return o if isinstance(o, Dual) else Dual(o, {})

    def _comb(self, o, fa, fb):
        """d = fa * self.d + fb * o.d over the union of slots (fa / fb None: factor 1)."""
        out = {}
        with np.errstate(all="ignore"):
            for s, da in self.d.items():
                out[s] = da if fa is None else fa * da
            for s, db in o.d.items():
                t = db if fb is None else fb * db
                out[s] = out[s] + t if s in out else t
        return out

    def __add__(self, o):
        o = Dual.lift(o)
        return Dual(self.v + o.v, self._comb(o, None, None))

    __radd__ = __add__

    def __neg__(self):
        return Dual(-self.v, {s: -d for s, d in self.d.items()})

    def __sub__(self, o):
        o = Dual.lift(o)
        return Dual(self.v - o.v, self._comb(o, None, -1.0))

    def __rsub__(self, o):
        return Dual.lift(o) - self

    def __mul__(self, o):
        o = Dual.lift(o)
        return Dual(self.v * o.v, self._comb(o, o.v, self.v))

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = Dual.lift(o)
        with np.errstate(all="ignore"):
            q = self.v / o.v
            return Dual(q, self._comb(o, 1.0 / o.v, -(q / o.v)))

    def __rtruediv__(self, o):
        return Dual.lift(o) / self

    def __pow__(self, o):
        return pow_(self, Dual.lift(o))


def _chain(v, f, a):
    with np.errstate(all="ignore"):
        return Dual(v, {s: f * d for s, d in a.d.items()})


def seed(x, params):
    """Dual variables for (x, *params), broadcast against each other."""
    arrs = np.broadcast_arrays(*[np.asarray(a, dtype=_F) for a in (x, *params)])
    return [Dual(a, {i: np.ones(a.shape)}) for i, a in enumerate(arrs)]


def log(a):
    with np.errstate(all="ignore"):
        return _chain(np.log(a.v), 1.0 / a.v, a)


def log1p(a):
    with np.errstate(all="ignore"):
        return _chain(np.log1p(a.v), 1.0 / (1.0 + a.v), a)


def exp(a):
    with np.errstate(all="ignore"):
        e = np.exp(a.v)
    return _chain(e, e, a)


def gammaln(a):
    with np.errstate(all="ignore"):
        return _chain(sp.gammaln(a.v), sp.digamma(a.v), a)


def sqr(a):
    return a * a


def abs_(a):
    return _chain(np.abs(a.v), np.sign(a.v), a)


def pow_(a, b):
    """PyTensor Pow.grad: [gz * y * x**(y-1), gz * log(x) * x**y]."""
    a, b = Dual.lift(a), Dual.lift(b)
    with np.errstate(all="ignore"):
        v = np.power(a.v, b.v)
        da = b.v * np.power(a.v, b.v - 1.0)
        db = np.log(a.v) * v
    return Dual(v, a._comb(b, da, db))


def xlogy0(a, y):
    """pytensor.tensor.xlogx.XlogY0: value 0 where a == 0; grad [gz * log(y), gz * a / y]."""
    a, y = Dual.lift(a), Dual.lift(y)
    with np.errstate(all="ignore"):
        ly = np.log(y.v)
        v = np.where(a.v == 0.0, 0.0, a.v * ly)
        return Dual(v, a._comb(y, ly, a.v / y.v))


def switch(cond, a, b):
    """pt.switch as a select: value and partials of the selected arm (an arm that lacks a slot contributes 0)."""
    a, b = Dual.lift(a), Dual.lift(b)
    cond = np.asarray(cond)
    d = {}
    for s in set(a.d) | set(b.d):
        d[s] = np.where(cond, a.d.get(s, 0.0), b.d.get(s, 0.0))
    return Dual(np.where(cond, a.v, b.v), d)


def betaln(a, b):
    """pytensor.tensor.special.betaln = gammaln(a) + gammaln(b) - gammaln(a + b)."""
    return gammaln(a) + gammaln(b) - gammaln(a + b)


def normal_logcdf(x, mu, sigma):
    """normal.py:70-76; d/dz = phi(z) / Phi(z) written as exp(-z^2/2 - log sqrt(2 pi) - logcdf)."""
    z = (x - mu) / sigma
    with np.errstate(all="ignore"):
        v = np.where(z.v < -1.0, np.log(sp.erfcx(-z.v / np.sqrt(2.0)) / 2.0) - 0.5 * z.v * z.v,
                     np.log1p(-sp.erfc(z.v / np.sqrt(2.0)) / 2.0))
        dz = np.exp(-0.5 * z.v * z.v - LOG_SQRT_2PI - v)
    return _chain(v, dz, z)


NEG_INF = -np.inf


# ------------------------------------------------------------------------------------ logpdf graphs
def normal(x, mu, sigma):  # normal.py:79-80
    return -0.5 * sqr((x - mu) / sigma) - log(sigma) - LOG_SQRT_2PI


def gamma(x, alpha, beta):  # gamma.py:77-82 (quirk Q2: xlogy0)
    return switch(x.v < 0, NEG_INF, xlogy0(alpha - 1.0, x) - beta * x + alpha * log(beta) - gammaln(alpha))


def beta(x, alpha, beta_):  # beta.py:102-108 (xlogy0(alpha+beta-1, 1) == 0 with zero partials)
    one_m = 1.0 - x
    return switch((x.v < 0) | (x.v > 1), NEG_INF,
                  xlogy0(alpha - 1.0, x) + xlogy0(beta_ - 1.0, one_m) - betaln(alpha, beta_))


def studentt(x, nu, mu, sigma):  # studentt.py:108-119 (quirk Q1: the nu > 1e5 arm is normal.logcdf)
    t = (gammaln((nu + 1.0) / 2.0) - gammaln(nu / 2.0) - 0.5 * log(nu * np.pi * sqr(sigma))
         - 0.5 * (nu + 1.0) * log1p(sqr(x - mu) / (sqr(sigma) * nu)))
    return switch(nu.v > 1e5, normal_logcdf(x, mu, sigma), t)


def poisson(x, mu):  # poisson.py:70-71
    return xlogy0(x, mu) - gammaln(x + 1.0) - mu


def binomial(x, n, p):  # binomial.py:66-75
    r = (gammaln(n + 1.0) - gammaln(x + 1.0) - gammaln(n - x + 1.0) + xlogy0(x, p) + (n - x) * log1p(-p))
    return switch((x.v < 0) | (x.v > n.v), NEG_INF, r)


def exponential(x, lam):  # exponential.py:71-72
    return switch(x.v < 0, NEG_INF, log(lam) - lam * x)


def lognormal(x, mu, sigma):  # lognormal.py:83-91
    lx = log(x)
    return switch(x.v <= 0, NEG_INF, -lx - log(sigma) - LOG_SQRT_2PI - 0.5 * sqr((lx - mu) / sigma))


def halfnormal(x, sigma):  # halfnormal.py:67-70
    return switch(x.v < 0, NEG_INF, (-0.22579135264472743236 - log(sigma)) - 0.5 * sqr(x / sigma))


def cauchy(x, alpha, beta_):  # cauchy.py:72-73
    return -LOG_PI - log(beta_) - log(1.0 + sqr((x - alpha) / beta_))


def halfcauchy(x, beta_):  # halfcauchy.py:72-75
    return switch(x.v < 0, NEG_INF, LOG_2 - log(np.pi * beta_) - log(1.0 + sqr(x / beta_)))


def laplace(x, mu, b):  # laplace.py:72-73
    return (-LOG_2 - abs_((x - mu) / b)) - log(b)


def logistic(x, mu, s):  # logistic.py:74-76
    z = (x - mu) / s
    return switch(np.isneginf(x.v), NEG_INF, (-log(s) - z) - 2.0 * log1p(exp(-z)))


def gumbel(x, mu, beta_):  # gumbel.py:74-76
    z = (x - mu) / beta_
    return (-z - exp(-z)) - log(beta_)


def weibull(x, alpha, beta_):  # weibull.py:82-87
    z = x / beta_
    return switch(x.v < 0, NEG_INF, log(alpha / beta_) + xlogy0(alpha - 1.0, z) - pow_(z, alpha))


def pareto(x, alpha, m):  # pareto.py:77-80
    return switch(x.v < m.v, NEG_INF, log(alpha) + alpha * log(m) - (alpha + 1.0) * log(x))


def uniform(x, lower, upper):  # uniform.py:78-80 (no x in the selected arm: d/dx = 0)
    return switch((x.v < lower.v) | (x.v > upper.v), NEG_INF, -log(upper - lower))


def triangular(x, lower, c, upper):  # triangular.py:91-97
    res = switch(x.v < c.v, log(2.0 * (x - lower) / ((upper - lower) * (c - lower))),
                 log(2.0 * (upper - x) / ((upper - lower) * (upper - c))))
    return switch((lower.v <= x.v) & (x.v <= upper.v), res, NEG_INF)


def kumaraswamy(x, a, b):  # kumaraswamy.py:80-85
    return switch((x.v <= 0) | (x.v >= 1), NEG_INF,
                  log(a * b) + xlogy0(a - 1.0, x) + xlogy0(b - 1.0, 1.0 - pow_(x, a)))


def moyal(x, mu, sigma):  # moyal.py:75-77
    z = (x - mu) / sigma
    return -0.5 * (z + exp(-z)) - log(sigma) - LOG_SQRT_2PI


def asymmetriclaplace(x, mu, b, kappa):  # asymmetriclaplace.py:88-95
    xn = (x - mu) / b
    kinv = 1.0 / kappa
    a = switch(xn.v >= 0, xn * (-kappa), xn * kinv)
    return (a - log(kappa + kinv)) - log(b)


def loglogistic(x, alpha, beta_):  # loglogistic.py:70-73
    z = x / alpha
    r = log(beta_) - log(alpha) + (beta_ - 1.0) * log(z) - 2.0 * log1p(pow_(z, beta_))
    return switch((x.v <= 0) | np.isinf(x.v), NEG_INF, r)


def logitnormal(x, mu, sigma):  # logitnormal.py:122-132 (logit = log(x) - log1p(-x), :11-12)
    lx, l1 = log(x), log1p(-x)
    z = ((lx - l1) - mu) / sigma
    return switch((x.v <= 0) | (x.v >= 1), NEG_INF, ((-0.5 * sqr(z) - log(sigma) - LOG_SQRT_2PI) - lx) - l1)


def inversegamma(x, alpha, beta_):  # inversegamma.py:76-81
    return switch(x.v <= 0, NEG_INF, (alpha * log(beta_) - gammaln(alpha) - (alpha + 1.0) * log(x)) - beta_ / x)


def chisquared(x, nu):  # chisquared.py:69-74
    h = nu / 2.0
    return switch(x.v < 0, NEG_INF, (xlogy0(h - 1.0, x) - x / 2.0) - gammaln(h) - (nu * LOG_2) / 2.0)


def scaledinversechisquared(x, nu, tau2):  # scaledinversechisquared.py:75-85
    h = nu / 2.0
    return switch(x.v <= 0, NEG_INF,
                  (-(log(x) * (h + 1.0)) - (h * tau2) / x) + (log(tau2) * h - gammaln(h)) + log(h) * h)


def halfstudentt(x, nu, sigma):  # halfstudentt.py:117-131
    t = switch(x.v < 0, NEG_INF,
               gammaln((nu + 1.0) / 2.0) - gammaln(nu / 2.0) - 0.5 * log(nu * np.pi * sqr(sigma))
               - 0.5 * (nu + 1.0) * log(1.0 + sqr(x / sigma) / nu) + LOG_2)
    return switch(nu.v > 1e5, halfnormal(x, sigma), t)


def betascaled(x, alpha, beta_, lower, upper):  # betascaled.py:106-112
    return switch((x.v < lower.v) | (x.v > upper.v), NEG_INF,
                  (xlogy0(alpha - 1.0, x - lower) + xlogy0(beta_ - 1.0, upper - x))
                  - (xlogy0(alpha + beta_ - 1.0, upper - lower) + betaln(alpha, beta_)))


GRAD_TABLE = {f.__name__: f for f in (
    normal, gamma, beta, studentt, poisson, binomial, exponential, lognormal, halfnormal, cauchy, halfcauchy,
    laplace, logistic, gumbel, weibull, pareto, uniform, triangular, kumaraswamy, moyal, asymmetriclaplace,
    loglogistic, logitnormal, inversegamma, chisquared, scaledinversechisquared, halfstudentt, betascaled)}


def logpdf_grad(name, x, *params):
    """(logpdf, [d/dx, d/dparam_1, ...]) elementwise on the broadcast shape of (x, *params)."""
    duals = seed(x, params)
    with np.errstate(all="ignore"):
        r = GRAD_TABLE[name](*duals)
    shape = duals[0].v.shape
    return np.broadcast_to(r.v, shape), [np.broadcast_to(r.d.get(i, 0.0), shape) for i in range(len(duals))]
