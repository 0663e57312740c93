// Kernel instantiations for studentt.py (see m355_elemwise.h).
#include "m355_elemwise.h"

int m355_launch_studentt(const m355::LaunchArgs& a) {
  return a.partials ? m355::launch_dist<M355_STUDENTT, true>(a) : m355::launch_dist<M355_STUDENTT, false>(a);
}
