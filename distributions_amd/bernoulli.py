"""bernoulli distribution: same function API as the reference's bernoulli.py (p), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists3.h); rvs returns int64."""

from ._family import export as _export

_export(globals(), "bernoulli", {'rvs': '41-42', 'cdf': '45-46', 'ppf': '49-51', 'sf': '54-55', 'isf': '58-59', 'pdf': '62-63', 'logpdf': '66-67', 'logcdf': '70-71', 'logsf': '74-75'})
