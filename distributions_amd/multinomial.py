"""Multinomial distribution: logpdf / pdf / rvs with the reference's signatures
(multinomial.py:35-58), evaluated by csrc/m355_catmult.hip."""

from ._catmult import multinomial_logpdf as _logpdf
from ._catmult import multinomial_rvs as _rvs


def logpdf(x, n, p):
    """multinomial.py:39-54 of the reference (-inf when any count is negative or sum(x) != n)."""
    return _logpdf(x, n, p)


def pdf(x, n, p):
    """multinomial.py:35-36 of the reference."""
    r = _logpdf(x, n, p)
    return r.exp() if hasattr(r, "is_cuda") else __import__("numpy").exp(r)


def rvs(n, p, size=None, random_state=None):
    """multinomial.py:57-58 of the reference; conditional-binomial Philox draws, rows sum to n."""
    return _rvs(n, p, size=size, random_state=random_state)
