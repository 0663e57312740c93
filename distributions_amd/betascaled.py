"""Betascaled distribution: same function API as the reference's betascaled.py (alpha, beta, lower, upper), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "betascaled", {'cdf': '71-72', 'isf': '75-76', 'pdf': '79-80', 'ppf': '83-85', 'sf': '88-89', 'rvs': '92-94', 'logcdf': '97-103', 'logpdf': '106-112', 'logsf': '115-121'})
