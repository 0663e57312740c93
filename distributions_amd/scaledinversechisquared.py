"""Scaledinversechisquared distribution: same function API as the reference's scaledinversechisquared.py (nu, tau2), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "scaledinversechisquared", {'cdf': '43-45', 'isf': '48-49', 'pdf': '52-53', 'ppf': '56-59', 'sf': '62-63', 'rvs': '66-67', 'logcdf': '70-72', 'logpdf': '75-85', 'logsf': '88-90'})
