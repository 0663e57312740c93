"""Studentt distribution: same function API as the reference's studentt.py, evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists.h)."""

from ._dispatch import elemwise as _elemwise
from ._dispatch import rvs as _rvs
from ._dispatch import sumlogp as _sumlogp


def logpdf(x, nu, mu, sigma):
    """studentt.py:108-119 of the reference."""
    return _elemwise("studentt", "logpdf", x, nu, mu, sigma)


def logcdf(x, nu, mu, sigma):
    """studentt.py:100-105 of the reference."""
    return _elemwise("studentt", "logcdf", x, nu, mu, sigma)


def logsf(x, nu, mu, sigma):
    """studentt.py:122-123 of the reference."""
    return _elemwise("studentt", "logsf", x, nu, mu, sigma)


def pdf(x, nu, mu, sigma):
    """studentt.py:75-76 of the reference."""
    return _elemwise("studentt", "pdf", x, nu, mu, sigma)


def cdf(x, nu, mu, sigma):
    """studentt.py:67-68 of the reference."""
    return _elemwise("studentt", "cdf", x, nu, mu, sigma)


def sf(x, nu, mu, sigma):
    """studentt.py:92-93 of the reference."""
    return _elemwise("studentt", "sf", x, nu, mu, sigma)


def ppf(q, nu, mu, sigma):
    """studentt.py:79-89 of the reference."""
    return _elemwise("studentt", "ppf", q, nu, mu, sigma)


def isf(q, nu, mu, sigma):
    """studentt.py:71-72 of the reference."""
    return _elemwise("studentt", "isf", q, nu, mu, sigma)


def sum_logpdf(x, nu, mu, sigma):
    """sum(logpdf(x, nu, mu, sigma)) fused in one pass (the reference spells it pt.sum(logpdf(...)))."""
    return _sumlogp("studentt", "logpdf", x, nu, mu, sigma)


def sum_logcdf(x, nu, mu, sigma):
    """sum(logcdf(x, ...)) fused in one pass (left-censored likelihoods; the reference spells it pt.sum(logcdf(...)))."""
    return _sumlogp("studentt", "logcdf", x, nu, mu, sigma)


def sum_logsf(x, nu, mu, sigma):
    """sum(logsf(x, ...)) fused in one pass (right-censored likelihoods; the reference spells it pt.sum(logsf(...)))."""
    return _sumlogp("studentt", "logsf", x, nu, mu, sigma)


def rvs(nu, mu, sigma, size=None, random_state=None):
    """studentt.py:96-97 of the reference; Philox counter-based draws, KS / chi-square equivalent to
    the reference sampler (not NumPy-stream-equal)."""
    return _rvs("studentt", (nu, mu, sigma,), size=size, random_state=random_state)


from ._family import export_grad as _export_grad  # noqa: E402

_export_grad(globals(), "studentt")

from ._family import export_moments as _export_moments  # noqa: E402

_export_moments(globals(), "studentt")
