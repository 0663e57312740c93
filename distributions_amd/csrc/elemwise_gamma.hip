// Kernel instantiations for gamma.py (see m355_elemwise.h).
#include "m355_elemwise.h"

int m355_launch_gamma(const m355::LaunchArgs& a) {
  return a.partials ? m355::launch_dist<M355_GAMMA, true>(a) : m355::launch_dist<M355_GAMMA, false>(a);
}
