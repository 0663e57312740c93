"""Weibull distribution: same function API as the reference's weibull.py (alpha, beta), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "weibull", {'cdf': '47-48', 'isf': '51-52', 'pdf': '55-56', 'ppf': '59-61', 'sf': '64-65', 'rvs': '68-69', 'logcdf': '72-79', 'logpdf': '82-87', 'logsf': '90-91'})
