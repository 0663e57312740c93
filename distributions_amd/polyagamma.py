"""polyagamma: same function API as the reference's polyagamma.py (h, z), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists4.h)."""

from ._family import export as _export

_export(globals(), "polyagamma", {'logpdf': '113-117', 'pdf': '120-121', 'cdf': '124-134', 'logcdf': '137-138', 'sf': '141-142', 'logsf': '145-146', 'ppf': '149-184', 'isf': '187-188', 'rvs': '191-208'})
