"""Gamma distribution: same function API as the reference's gamma.py, evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists.h)."""

from ._dispatch import elemwise as _elemwise
from ._dispatch import rvs as _rvs
from ._dispatch import sumlogp as _sumlogp


def logpdf(x, alpha, beta):
    """gamma.py:77-82 of the reference."""
    return _elemwise("gamma", "logpdf", x, alpha, beta)


def logcdf(x, alpha, beta):
    """gamma.py:65-74 of the reference."""
    return _elemwise("gamma", "logcdf", x, alpha, beta)


def logsf(x, alpha, beta):
    """gamma.py:85-94 of the reference."""
    return _elemwise("gamma", "logsf", x, alpha, beta)


def pdf(x, alpha, beta):
    """gamma.py:44-45 of the reference."""
    return _elemwise("gamma", "pdf", x, alpha, beta)


def cdf(x, alpha, beta):
    """gamma.py:40-41 of the reference."""
    return _elemwise("gamma", "cdf", x, alpha, beta)


def sf(x, alpha, beta):
    """gamma.py:53-54 of the reference."""
    return _elemwise("gamma", "sf", x, alpha, beta)


def ppf(q, alpha, beta):
    """gamma.py:48-50 of the reference."""
    return _elemwise("gamma", "ppf", q, alpha, beta)


def isf(q, alpha, beta):
    """gamma.py:57-58 of the reference."""
    return _elemwise("gamma", "isf", q, alpha, beta)


def sum_logpdf(x, alpha, beta):
    """sum(logpdf(x, alpha, beta)) fused in one pass (the reference spells it pt.sum(logpdf(...)))."""
    return _sumlogp("gamma", "logpdf", x, alpha, beta)


def sum_logcdf(x, alpha, beta):
    """sum(logcdf(x, ...)) fused in one pass (left-censored likelihoods; the reference spells it pt.sum(logcdf(...)))."""
    return _sumlogp("gamma", "logcdf", x, alpha, beta)


def sum_logsf(x, alpha, beta):
    """sum(logsf(x, ...)) fused in one pass (right-censored likelihoods; the reference spells it pt.sum(logsf(...)))."""
    return _sumlogp("gamma", "logsf", x, alpha, beta)


def rvs(alpha, beta, size=None, random_state=None):
    """gamma.py:61-62 (shape alpha, scale 1/beta) of the reference; Philox counter-based draws, KS / chi-square equivalent to
    the reference sampler (not NumPy-stream-equal)."""
    return _rvs("gamma", (alpha, beta,), size=size, random_state=random_state)


from ._family import export_grad as _export_grad  # noqa: E402

_export_grad(globals(), "gamma")

from ._family import export_moments as _export_moments  # noqa: E402

_export_moments(globals(), "gamma")
