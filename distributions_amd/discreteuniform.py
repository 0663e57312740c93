"""discreteuniform distribution: same function API as the reference's discreteuniform.py (lower, upper), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists3.h); rvs returns int64."""

from ._family import export as _export

_export(globals(), "discreteuniform", {'pdf': '41-42', 'cdf': '45-47', 'ppf': '50-53', 'sf': '56-57', 'isf': '60-61', 'rvs': '64-65', 'logpdf': '68-70', 'logcdf': '73-74', 'logsf': '77-78'})
