// Kernel instantiations for zi_binomial.py (see m355_elemwise.h).
#include "m355_elemwise.h"

M355_DEFINE_LAUNCH(M355_ZI_BINOMIAL, zi_binomial)
