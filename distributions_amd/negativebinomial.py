"""negativebinomial distribution: same function API as the reference's negativebinomial.py (n, p), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists3.h); rvs returns int64."""

from ._family import export as _export

_export(globals(), "negativebinomial", {'pdf': '42-43', 'cdf': '46-48', 'ppf': '51-53', 'sf': '56-59', 'isf': '62-63', 'rvs': '66-67', 'logpdf': '70-75', 'logcdf': '78-79', 'logsf': '82-83'})
