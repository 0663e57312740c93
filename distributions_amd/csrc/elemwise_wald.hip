// Kernel instantiations for wald.py (see m355_elemwise.h).
#include "m355_elemwise.h"

M355_DEFINE_LAUNCH(M355_WALD, wald)
