"""Halfcauchy distribution: same function API as the reference's halfcauchy.py (beta), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "halfcauchy", {'cdf': '38-39', 'logcdf': '42-47', 'isf': '50-51', 'pdf': '54-55', 'ppf': '58-60', 'sf': '63-64', 'rvs': '67-69', 'logpdf': '72-75', 'logsf': '78-79'})
