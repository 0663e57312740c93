"""Exponential distribution: same function API as the reference's exponential.py (lam), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "exponential", {'cdf': '41-43', 'isf': '46-47', 'pdf': '50-51', 'ppf': '54-56', 'sf': '59-60', 'rvs': '63-64', 'logcdf': '67-68', 'logpdf': '71-72', 'logsf': '75-76'})
