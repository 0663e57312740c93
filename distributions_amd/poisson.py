"""Poisson distribution: same function API as the reference's poisson.py, evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists.h)."""

from ._dispatch import elemwise as _elemwise
from ._dispatch import rvs as _rvs
from ._dispatch import sumlogp as _sumlogp


def logpdf(x, mu):
    """poisson.py:70-71 of the reference."""
    return _elemwise("poisson", "logpdf", x, mu)


def logcdf(x, mu):
    """poisson.py:74-75 of the reference."""
    return _elemwise("poisson", "logcdf", x, mu)


def logsf(x, mu):
    """poisson.py:78-79 of the reference."""
    return _elemwise("poisson", "logsf", x, mu)


def pdf(x, mu):
    """poisson.py:45-46 of the reference."""
    return _elemwise("poisson", "pdf", x, mu)


def cdf(x, mu):
    """poisson.py:49-50 of the reference."""
    return _elemwise("poisson", "cdf", x, mu)


def sf(x, mu):
    """poisson.py:58-59 of the reference."""
    return _elemwise("poisson", "sf", x, mu)


def ppf(q, mu):
    """poisson.py:53-55 of the reference (find_ppf_discrete, optimization.py:102-129, run per element)."""
    return _elemwise("poisson", "ppf", q, mu)


def isf(q, mu):
    """poisson.py:62-63 of the reference."""
    return _elemwise("poisson", "isf", q, mu)


def sum_logpdf(x, mu):
    """sum(logpdf(x, mu)) fused in one pass (the reference spells it pt.sum(logpdf(...)))."""
    return _sumlogp("poisson", "logpdf", x, mu)


def sum_logcdf(x, mu):
    """sum(logcdf(x, ...)) fused in one pass (left-censored likelihoods; the reference spells it pt.sum(logcdf(...)))."""
    return _sumlogp("poisson", "logcdf", x, mu)


def sum_logsf(x, mu):
    """sum(logsf(x, ...)) fused in one pass (right-censored likelihoods; the reference spells it pt.sum(logsf(...)))."""
    return _sumlogp("poisson", "logsf", x, mu)


def rvs(mu, size=None, random_state=None):
    """poisson.py:66-67 (int64) of the reference; Philox counter-based draws, KS / chi-square equivalent to
    the reference sampler (not NumPy-stream-equal)."""
    return _rvs("poisson", (mu,), size=size, random_state=random_state)


from ._family import export_grad as _export_grad  # noqa: E402

_export_grad(globals(), "poisson")

from ._family import export_moments as _export_moments  # noqa: E402

_export_moments(globals(), "poisson")
