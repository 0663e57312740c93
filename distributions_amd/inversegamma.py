"""Inversegamma distribution: same function API as the reference's inversegamma.py (alpha, beta), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "inversegamma", {'cdf': '48-49', 'isf': '52-53', 'pdf': '56-57', 'ppf': '60-61', 'sf': '64-65', 'rvs': '68-69', 'logcdf': '72-73', 'logpdf': '76-81', 'logsf': '84-85'})
