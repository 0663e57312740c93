"""Categorical distribution: same function API as the reference's categorical.py (density, CDF and
quantile functions of one probability vector p), evaluated by csrc/m355_catmult.hip.  Integer
observations are compared as integers: the support / index logic is exact."""

from ._catmult import categorical as _cat
from ._catmult import categorical_rvs as _rvs


def pdf(x, p):
    """categorical.py:60-61 of the reference."""
    return _cat("pdf", x, p)


def logpdf(x, p):
    """categorical.py:64-74 of the reference."""
    return _cat("logpdf", x, p)


def cdf(x, p):
    """categorical.py:77-85 of the reference."""
    return _cat("cdf", x, p)


def logcdf(x, p):
    """categorical.py:88-89 of the reference."""
    return _cat("logcdf", x, p)


def sf(x, p):
    """categorical.py:92-93 of the reference."""
    return _cat("sf", x, p)


def logsf(x, p):
    """categorical.py:96-97 of the reference."""
    return _cat("logsf", x, p)


def ppf(q, p):
    """categorical.py:100-108 of the reference."""
    return _cat("ppf", q, p)


def isf(q, p):
    """categorical.py:111-112 of the reference."""
    return _cat("isf", q, p)


def rvs(p, size=None, random_state=None):
    """categorical.py:115-117 of the reference; Philox inverse-CDF draws (int64), statistically
    equivalent to the reference sampler."""
    return _rvs(p, size=size, random_state=random_state)
