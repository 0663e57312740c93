"""Lognormal distribution: same function API as the reference's lognormal.py (mu, sigma), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "lognormal", {'cdf': '45-47', 'isf': '50-51', 'pdf': '54-55', 'ppf': '58-59', 'sf': '62-63', 'rvs': '66-67', 'logcdf': '70-80', 'logpdf': '83-91', 'logsf': '94-104'})
