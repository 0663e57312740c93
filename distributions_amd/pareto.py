"""Pareto distribution: same function API as the reference's pareto.py (alpha, m), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "pareto", {'cdf': '47-48', 'pdf': '51-52', 'ppf': '55-57', 'sf': '60-61', 'isf': '64-65', 'rvs': '68-70', 'logcdf': '73-74', 'logpdf': '77-80', 'logsf': '83-84'})
