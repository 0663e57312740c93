// Kernel instantiations for beta.py (see m355_elemwise.h).
#include "m355_elemwise.h"

int m355_launch_beta(const m355::LaunchArgs& a) {
  return a.partials ? m355::launch_dist<M355_BETA, true>(a) : m355::launch_dist<M355_BETA, false>(a);
}
