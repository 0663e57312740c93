"""Multivariate normal: logpdf / pdf with the reference's signature (mvnormal.py:76-83), evaluated
by the batched Cholesky + forward-substitution kernels of csrc/m355_mvn.hip."""

from ._mv import mv_logpdf as _mv_logpdf


def logpdf(x, mu, cov):
    """mvnormal.py:80-83 (quaddist_chol :17-37; posdef is ignored: a non-PD cov gives NaN)."""
    return _mv_logpdf(0, x, mu, cov)


def pdf(x, mu, cov):
    """mvnormal.py:76-77."""
    r = logpdf(x, mu, cov)
    return r.exp() if hasattr(r, "exp") and not hasattr(r, "__array_interface__") else __import__("numpy").exp(r)
