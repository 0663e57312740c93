"""Halfnormal distribution: same function API as the reference's halfnormal.py (sigma), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "halfnormal", {'cdf': '38-39', 'logcdf': '42-43', 'isf': '46-47', 'pdf': '50-51', 'ppf': '54-56', 'sf': '59-60', 'rvs': '63-64', 'logpdf': '67-70', 'logsf': '73-74'})
