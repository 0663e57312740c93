// Kernel instantiations for skew_studentt.py (see m355_elemwise.h).
#include "m355_elemwise.h"

M355_DEFINE_LAUNCH(M355_SKEW_STUDENTT, skew_studentt)
