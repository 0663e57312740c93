// Kernel instantiations for normal.py (see m355_elemwise.h).
#include "m355_elemwise.h"

int m355_launch_normal(const m355::LaunchArgs& a) {
  return a.partials ? m355::launch_dist<M355_NORMAL, true>(a) : m355::launch_dist<M355_NORMAL, false>(a);
}
