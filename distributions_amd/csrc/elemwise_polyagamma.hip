// Kernel instantiations for polyagamma.py (see m355_elemwise.h).
#include "m355_elemwise.h"

M355_DEFINE_LAUNCH(M355_POLYAGAMMA, polyagamma)
