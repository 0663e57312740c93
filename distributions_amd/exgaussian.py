"""exgaussian: same function API as the reference's exgaussian.py (mu, sigma, nu), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists4.h)."""

from ._family import export as _export

_export(globals(), "exgaussian", {'cdf': '50-51', 'isf': '54-55', 'pdf': '58-59', 'ppf': '62-63', 'sf': '66-67', 'rvs': '70-73', 'logcdf': '76-98', 'logpdf': '101-111', 'logsf': '114-115'})
