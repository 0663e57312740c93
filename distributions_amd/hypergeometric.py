"""hypergeometric distribution: same function API as the reference's hypergeometric.py (N, k, n), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists3.h); rvs returns int64."""

from ._family import export as _export

_export(globals(), "hypergeometric", {'pdf': '63-64', 'logpdf': '67-73', 'cdf': '76-90', 'logcdf': '93-94', 'sf': '97-98', 'logsf': '101-102', 'ppf': '105-108', 'isf': '111-112', 'rvs': '115-116'})
