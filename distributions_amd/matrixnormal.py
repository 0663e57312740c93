"""Matrix normal: logpdf / pdf with the reference's signature (matrixnormal.py:68-98)."""

from ._mv import matrixnormal_logpdf as _logpdf


def logpdf(X, mu, rowcov, colcov):
    """matrixnormal.py:68-94."""
    return _logpdf(X, mu, rowcov, colcov)


def pdf(X, mu, rowcov, colcov):
    """matrixnormal.py:97-98."""
    r = logpdf(X, mu, rowcov, colcov)
    return r.exp() if hasattr(r, "exp") and not hasattr(r, "__array_interface__") else __import__("numpy").exp(r)
