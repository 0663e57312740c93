"""Triangular distribution: same function API as the reference's triangular.py (lower, c, upper), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "triangular", {'pdf': '44-50', 'cdf': '53-67', 'ppf': '70-75', 'sf': '78-79', 'isf': '82-83', 'rvs': '86-88', 'logpdf': '91-97', 'logcdf': '100-113', 'logsf': '116-129'})
