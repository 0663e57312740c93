"""Binomial distribution: same function API as the reference's binomial.py, evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists.h)."""

from ._dispatch import elemwise as _elemwise
from ._dispatch import rvs as _rvs
from ._dispatch import sumlogp as _sumlogp


def logpdf(x, n, p):
    """binomial.py:66-75 of the reference."""
    return _elemwise("binomial", "logpdf", x, n, p)


def logcdf(x, n, p):
    """binomial.py:78-79 of the reference."""
    return _elemwise("binomial", "logcdf", x, n, p)


def logsf(x, n, p):
    """binomial.py:82-83 of the reference."""
    return _elemwise("binomial", "logsf", x, n, p)


def pdf(x, n, p):
    """binomial.py:41-42 of the reference."""
    return _elemwise("binomial", "pdf", x, n, p)


def cdf(x, n, p):
    """binomial.py:45-46 of the reference."""
    return _elemwise("binomial", "cdf", x, n, p)


def sf(x, n, p):
    """binomial.py:54-55 of the reference."""
    return _elemwise("binomial", "sf", x, n, p)


def ppf(q, n, p):
    """binomial.py:49-51 of the reference (find_ppf_discrete, optimization.py:102-129, run per element)."""
    return _elemwise("binomial", "ppf", q, n, p)


def isf(q, n, p):
    """binomial.py:58-59 of the reference."""
    return _elemwise("binomial", "isf", q, n, p)


def sum_logpdf(x, n, p):
    """sum(logpdf(x, n, p)) fused in one pass (the reference spells it pt.sum(logpdf(...)))."""
    return _sumlogp("binomial", "logpdf", x, n, p)


def sum_logcdf(x, n, p):
    """sum(logcdf(x, ...)) fused in one pass (left-censored likelihoods; the reference spells it pt.sum(logcdf(...)))."""
    return _sumlogp("binomial", "logcdf", x, n, p)


def sum_logsf(x, n, p):
    """sum(logsf(x, ...)) fused in one pass (right-censored likelihoods; the reference spells it pt.sum(logsf(...)))."""
    return _sumlogp("binomial", "logsf", x, n, p)


def rvs(n, p, size=None, random_state=None):
    """binomial.py:62-63 (int64) of the reference; Philox counter-based draws, KS / chi-square equivalent to
    the reference sampler (not NumPy-stream-equal)."""
    return _rvs("binomial", (n, p,), size=size, random_state=random_state)


from ._family import export_grad as _export_grad  # noqa: E402

_export_grad(globals(), "binomial")

from ._family import export_moments as _export_moments  # noqa: E402

_export_moments(globals(), "binomial")
