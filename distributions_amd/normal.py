"""Normal distribution: same function API as the reference's normal.py, evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists.h)."""

from ._dispatch import elemwise as _elemwise
from ._dispatch import rvs as _rvs
from ._dispatch import sumlogp as _sumlogp


def logpdf(x, mu, sigma):
    """normal.py:79-80 of the reference."""
    return _elemwise("normal", "logpdf", x, mu, sigma)


def logcdf(x, mu, sigma):
    """normal.py:70-76 of the reference."""
    return _elemwise("normal", "logcdf", x, mu, sigma)


def logsf(x, mu, sigma):
    """normal.py:83-84 of the reference."""
    return _elemwise("normal", "logsf", x, mu, sigma)


def pdf(x, mu, sigma):
    """normal.py:54-55 of the reference."""
    return _elemwise("normal", "pdf", x, mu, sigma)


def cdf(x, mu, sigma):
    """normal.py:46-47 of the reference."""
    return _elemwise("normal", "cdf", x, mu, sigma)


def sf(x, mu, sigma):
    """normal.py:62-63 of the reference."""
    return _elemwise("normal", "sf", x, mu, sigma)


def ppf(q, mu, sigma):
    """normal.py:58-59 of the reference."""
    return _elemwise("normal", "ppf", q, mu, sigma)


def isf(q, mu, sigma):
    """normal.py:50-51 of the reference."""
    return _elemwise("normal", "isf", q, mu, sigma)


def sum_logpdf(x, mu, sigma):
    """sum(logpdf(x, mu, sigma)) fused in one pass (the reference spells it pt.sum(logpdf(...)))."""
    return _sumlogp("normal", "logpdf", x, mu, sigma)


def sum_logcdf(x, mu, sigma):
    """sum(logcdf(x, ...)) fused in one pass (left-censored likelihoods; the reference spells it pt.sum(logcdf(...)))."""
    return _sumlogp("normal", "logcdf", x, mu, sigma)


def sum_logsf(x, mu, sigma):
    """sum(logsf(x, ...)) fused in one pass (right-censored likelihoods; the reference spells it pt.sum(logsf(...)))."""
    return _sumlogp("normal", "logsf", x, mu, sigma)


def rvs(mu, sigma, size=None, random_state=None):
    """normal.py:66-67 of the reference; Philox counter-based draws, KS / chi-square equivalent to
    the reference sampler (not NumPy-stream-equal)."""
    return _rvs("normal", (mu, sigma,), size=size, random_state=random_state)


from ._family import export_grad as _export_grad  # noqa: E402

_export_grad(globals(), "normal")

from ._family import export_moments as _export_moments  # noqa: E402

_export_moments(globals(), "normal")
