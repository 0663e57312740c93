"""zi_poisson distribution: same function API as the reference's zi_poisson.py (psi, mu), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists3.h); rvs returns int64."""

from ._family import export as _export

_export(globals(), "zi_poisson", {'pdf': '62-69', 'logpdf': '72-90', 'cdf': '93-96', 'ppf': '99-101', 'sf': '104-107', 'isf': '110-111', 'rvs': '114-117', 'logcdf': '120-127', 'logsf': '130-131'})
