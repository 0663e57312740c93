"""vonmises: same function API as the reference's vonmises.py (mu, kappa), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists4.h)."""

from ._family import export as _export

_export(globals(), "vonmises", {'pdf': '41-42', 'logpdf': '45-46', 'rvs': '49-50', 'cdf': '53-54', 'logcdf': '57-58', 'sf': '61-62', 'logsf': '65-66', 'isf': '69-70', 'ppf': '73-74'})
