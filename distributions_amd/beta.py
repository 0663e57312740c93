"""Beta distribution: same function API as the reference's beta.py, evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists.h)."""

from ._dispatch import elemwise as _elemwise
from ._dispatch import rvs as _rvs
from ._dispatch import sumlogp as _sumlogp


def logpdf(x, alpha, beta):
    """beta.py:102-108 of the reference."""
    return _elemwise("beta", "logpdf", x, alpha, beta)


def logcdf(x, alpha, beta):
    """beta.py:90-99 of the reference."""
    return _elemwise("beta", "logcdf", x, alpha, beta)


def logsf(x, alpha, beta):
    """beta.py:111-120 of the reference."""
    return _elemwise("beta", "logsf", x, alpha, beta)


def pdf(x, alpha, beta):
    """beta.py:74-75 of the reference."""
    return _elemwise("beta", "pdf", x, alpha, beta)


def cdf(x, alpha, beta):
    """beta.py:66-67 of the reference."""
    return _elemwise("beta", "cdf", x, alpha, beta)


def sf(x, alpha, beta):
    """beta.py:82-83 of the reference."""
    return _elemwise("beta", "sf", x, alpha, beta)


def ppf(q, alpha, beta):
    """beta.py:78-79 of the reference."""
    return _elemwise("beta", "ppf", q, alpha, beta)


def isf(q, alpha, beta):
    """beta.py:70-71 of the reference."""
    return _elemwise("beta", "isf", q, alpha, beta)


def sum_logpdf(x, alpha, beta):
    """sum(logpdf(x, alpha, beta)) fused in one pass (the reference spells it pt.sum(logpdf(...)))."""
    return _sumlogp("beta", "logpdf", x, alpha, beta)


def sum_logcdf(x, alpha, beta):
    """sum(logcdf(x, ...)) fused in one pass (left-censored likelihoods; the reference spells it pt.sum(logcdf(...)))."""
    return _sumlogp("beta", "logcdf", x, alpha, beta)


def sum_logsf(x, alpha, beta):
    """sum(logsf(x, ...)) fused in one pass (right-censored likelihoods; the reference spells it pt.sum(logsf(...)))."""
    return _sumlogp("beta", "logsf", x, alpha, beta)


def rvs(alpha, beta, size=None, random_state=None):
    """beta.py:86-87 of the reference; Philox counter-based draws, KS / chi-square equivalent to
    the reference sampler (not NumPy-stream-equal)."""
    return _rvs("beta", (alpha, beta,), size=size, random_state=random_state)


from ._family import export_grad as _export_grad  # noqa: E402

_export_grad(globals(), "beta")

from ._family import export_moments as _export_moments  # noqa: E402

_export_moments(globals(), "beta")
