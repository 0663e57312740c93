"""Uniform distribution: same function API as the reference's uniform.py (lower, upper), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "uniform", {'cdf': '41-46', 'pdf': '49-50', 'ppf': '53-55', 'sf': '58-59', 'isf': '62-63', 'rvs': '66-67', 'logcdf': '70-75', 'logpdf': '78-80', 'logsf': '83-84'})
