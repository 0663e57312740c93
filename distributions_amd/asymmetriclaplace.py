"""Asymmetriclaplace distribution: same function API as the reference's asymmetriclaplace.py (mu, b, kappa), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "asymmetriclaplace", {'cdf': '43-52', 'ppf': '55-65', 'pdf': '68-69', 'sf': '72-73', 'isf': '76-77', 'rvs': '80-85', 'logpdf': '88-95', 'logcdf': '98-99', 'logsf': '102-103'})
