// Kernel instantiations for zi_poisson.py (see m355_elemwise.h).
#include "m355_elemwise.h"

M355_DEFINE_LAUNCH(M355_ZI_POISSON, zi_poisson)
