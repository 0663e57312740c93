"""geometric distribution: same function API as the reference's geometric.py (p), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists3.h); rvs returns int64."""

from ._family import export as _export

_export(globals(), "geometric", {'pdf': '38-39', 'cdf': '42-45', 'ppf': '48-52', 'sf': '55-56', 'isf': '59-60', 'rvs': '63-64', 'logpdf': '67-68', 'logcdf': '71-72', 'logsf': '75-76'})
