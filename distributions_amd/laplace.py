"""Laplace distribution: same function API as the reference's laplace.py (mu, b), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "laplace", {'pdf': '46-47', 'cdf': '50-52', 'sf': '55-56', 'ppf': '59-61', 'isf': '64-65', 'rvs': '68-69', 'logpdf': '72-73', 'logcdf': '76-86', 'logsf': '89-90'})
