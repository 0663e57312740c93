"""(dist, fn) pairs that have an elementwise kernel (mirrors m355::supported in csrc/m355_dists.h)."""

from ._lib import DIST, FN

SUPPORTED = {(d, f) for d in DIST for f in FN}
