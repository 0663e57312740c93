"""skewnormal: same function API as the reference's skewnormal.py (mu, sigma, alpha), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists4.h)."""

from ._family import export as _export

_export(globals(), "skewnormal", {'cdf': '61-62', 'logcdf': '65-66', 'isf': '69-70', 'pdf': '73-74', 'ppf': '77-80', 'sf': '83-84', 'rvs': '87-92', 'logpdf': '95-100', 'logsf': '103-104'})
