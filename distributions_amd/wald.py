"""wald: same function API as the reference's wald.py (mu, lam), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists4.h)."""

from ._family import export as _export

_export(globals(), "wald", {'cdf': '40-47', 'isf': '50-51', 'pdf': '54-55', 'ppf': '58-60', 'sf': '63-64', 'rvs': '67-68', 'logcdf': '73-74', 'logpdf': '77-82', 'logsf': '85-86'})
