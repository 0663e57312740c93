"""truncatednormal: same function API as the reference's truncatednormal.py (mu, sigma, lower, upper), evaluated by the HIP
elementwise kernels (csrc/m355_elemwise.h, bodies in csrc/m355_dists4.h)."""

from ._family import export as _export

_export(globals(), "truncatednormal", {'pdf': '168-169', 'logpdf': '172-183', 'cdf': '186-206', 'logcdf': '209-222', 'sf': '225-226', 'logsf': '229-242', 'ppf': '245-266', 'isf': '269-270', 'rvs': '273-275'})
