"""Chisquared distribution: same function API as the reference's chisquared.py (nu), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "chisquared", {'cdf': '40-41', 'ppf': '44-46', 'pdf': '49-50', 'rvs': '53-54', 'logcdf': '57-66', 'logpdf': '69-74', 'sf': '77-78', 'isf': '81-82', 'logsf': '85-94'})
