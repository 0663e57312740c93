// Kernel instantiations for poisson.py (see m355_elemwise.h).
#include "m355_elemwise.h"

int m355_launch_poisson(const m355::LaunchArgs& a) {
  return a.partials ? m355::launch_dist<M355_POISSON, true>(a) : m355::launch_dist<M355_POISSON, false>(a);
}
