"""Logistic distribution: same function API as the reference's logistic.py (mu, s), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "logistic", {'cdf': '46-47', 'isf': '50-51', 'pdf': '54-55', 'ppf': '58-59', 'sf': '62-63', 'rvs': '66-67', 'logcdf': '70-71', 'logpdf': '74-76', 'logsf': '79-80'})
