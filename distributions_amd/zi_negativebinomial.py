"""zi_negativebinomial distribution: same function API as the reference's zi_negativebinomial.py (psi, n, p), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists3.h); rvs returns int64."""

from ._family import export as _export

_export(globals(), "zi_negativebinomial", {'pdf': '82-88', 'logpdf': '91-109', 'cdf': '112-115', 'ppf': '118-120', 'sf': '123-126', 'isf': '129-130', 'rvs': '133-136', 'logcdf': '139-146', 'logsf': '149-150'})
