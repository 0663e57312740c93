"""Loglogistic distribution: same function API as the reference's loglogistic.py (alpha, beta), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "loglogistic", {'pdf': '66-67', 'logpdf': '70-73', 'cdf': '76-79', 'logcdf': '82-88', 'sf': '91-97', 'logsf': '100-106', 'ppf': '109-111', 'isf': '114-115', 'rvs': '118-120'})
