"""Moyal distribution: same function API as the reference's moyal.py (mu, sigma), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists2.h)."""

from ._family import export as _export

_export(globals(), "moyal", {'cdf': '43-45', 'pdf': '48-49', 'ppf': '52-54', 'sf': '57-58', 'isf': '61-62', 'rvs': '65-67', 'logcdf': '70-72', 'logpdf': '75-77', 'logsf': '80-81'})
