"""betabinomial distribution: same function API as the reference's betabinomial.py (n, alpha, beta), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists3.h); rvs returns int64."""

from ._family import export as _export

_export(globals(), "betabinomial", {'pdf': '83-84', 'cdf': '87-95', 'ppf': '98-100', 'sf': '103-104', 'isf': '107-108', 'rvs': '111-112', 'logcdf': '115-116', 'logsf': '119-120', 'logpdf': '123-138'})
