"""Multivariate Student t: logpdf / pdf with the reference's signature (mvstudentt.py:54-68)."""

from ._mv import mv_logpdf as _mv_logpdf


def logpdf(x, nu, mu, cov):
    """mvstudentt.py:58-68 (nu scalar; the outer switch on x < -inf | x > inf is always false)."""
    return _mv_logpdf(1, x, mu, cov, nu=float(nu))


def pdf(x, nu, mu, cov):
    """mvstudentt.py:54-55."""
    r = logpdf(x, nu, mu, cov)
    return r.exp() if hasattr(r, "exp") and not hasattr(r, "__array_interface__") else __import__("numpy").exp(r)
