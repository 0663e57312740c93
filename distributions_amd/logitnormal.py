"""Logitnormal distribution: same function API as the reference's logitnormal.py (mu, sigma), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "logitnormal", {'pdf': '118-119', 'logpdf': '122-132', 'cdf': '135-138', 'logcdf': '141-156', 'sf': '159-160', 'logsf': '163-178', 'ppf': '181-182', 'isf': '185-186', 'rvs': '189-190'})
