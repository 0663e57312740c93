"""MI355X-native evaluation path for pytensor_distributions' per-distribution functions.

Same module-per-distribution API as the reference (`from distributions_amd import normal;
normal.logpdf(x, mu, sigma)`), evaluated eagerly by hand-written HIP kernels (gfx950) behind the
C-ABI of include/m355_dist.h.  There is no CPU fallback: without the built library and a GPU every
call raises.
"""

from ._lib import M355Error, lib

__version__ = "0.1.0"
__all__ = ["lib", "M355Error", "__version__"]
