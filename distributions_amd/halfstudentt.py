"""Halfstudentt distribution: same function API as the reference's halfstudentt.py (nu, sigma), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "halfstudentt", {'cdf': '76-83', 'isf': '86-87', 'pdf': '90-91', 'ppf': '94-101', 'sf': '104-105', 'rvs': '108-110', 'logcdf': '113-114', 'logpdf': '117-131', 'logsf': '134-135'})
