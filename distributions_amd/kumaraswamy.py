"""Kumaraswamy distribution: same function API as the reference's kumaraswamy.py (a, b), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "kumaraswamy", {'cdf': '53-55', 'pdf': '58-59', 'ppf': '62-64', 'sf': '67-68', 'isf': '71-72', 'rvs': '75-77', 'logpdf': '80-85', 'logsf': '88-89', 'logcdf': '92-93'})
