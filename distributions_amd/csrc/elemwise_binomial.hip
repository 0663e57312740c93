// Kernel instantiations for binomial.py (see m355_elemwise.h).
#include "m355_elemwise.h"

int m355_launch_binomial(const m355::LaunchArgs& a) {
  return a.partials ? m355::launch_dist<M355_BINOMIAL, true>(a) : m355::launch_dist<M355_BINOMIAL, false>(a);
}
