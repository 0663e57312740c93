"""skew_studentt: same function API as the reference's skew_studentt.py (a, b, mu, sigma), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists4.h)."""

from ._family import export as _export

_export(globals(), "skew_studentt", {'pdf': '109-110', 'logpdf': '113-120', 'cdf': '123-129', 'logcdf': '132-138', 'sf': '141-143', 'logsf': '146-147', 'ppf': '150-157', 'isf': '160-161', 'rvs': '164-166'})
