"""Gumbel distribution: same function API as the reference's gumbel.py (mu, beta), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "gumbel", {'cdf': '44-46', 'isf': '49-50', 'pdf': '53-54', 'ppf': '57-59', 'sf': '62-63', 'rvs': '66-67', 'logcdf': '70-71', 'logpdf': '74-76', 'logsf': '79-80'})
