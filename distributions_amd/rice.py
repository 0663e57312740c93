"""rice: same function API as the reference's rice.py (nu, sigma), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists4.h)."""

from ._family import export as _export

_export(globals(), "rice", {'pdf': '61-62', 'logpdf': '65-78', 'cdf': '81-83', 'logcdf': '86-91', 'sf': '94-95', 'logsf': '98-99', 'ppf': '102-103', 'isf': '106-107', 'rvs': '110-113'})
