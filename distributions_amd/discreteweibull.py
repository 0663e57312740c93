"""discreteweibull distribution: same function API as the reference's discreteweibull.py (q, beta), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists3.h); rvs returns int64."""

from ._family import export as _export

_export(globals(), "discreteweibull", {'cdf': '49-50', 'isf': '53-54', 'pdf': '57-62', 'ppf': '65-67', 'sf': '70-75', 'rvs': '78-79', 'logcdf': '82-87', 'logpdf': '90-95', 'logsf': '98-103'})
