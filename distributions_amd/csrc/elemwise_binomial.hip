// Kernel instantiations for binomial.py (see m355_elemwise.h).
#include "m355_elemwise.h"

M355_DEFINE_LAUNCH(M355_BINOMIAL, binomial)
