"""Cauchy distribution: same function API as the reference's cauchy.py (alpha, beta), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "cauchy", {'cdf': '44-45', 'isf': '48-49', 'pdf': '52-53', 'ppf': '56-57', 'sf': '60-61', 'rvs': '64-65', 'logcdf': '68-69', 'logpdf': '72-73', 'logsf': '76-78'})
