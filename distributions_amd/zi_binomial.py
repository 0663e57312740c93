"""zi_binomial distribution: same function API as the reference's zi_binomial.py (psi, n, p), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists3.h); rvs returns int64."""

from ._family import export as _export

_export(globals(), "zi_binomial", {'pdf': '75-76', 'logpdf': '79-90', 'cdf': '93-96', 'ppf': '99-101', 'sf': '104-105', 'isf': '108-109', 'rvs': '112-115', 'logcdf': '118-125', 'logsf': '128-129'})
