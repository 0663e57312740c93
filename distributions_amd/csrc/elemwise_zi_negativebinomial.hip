// Kernel instantiations for zi_negativebinomial.py (see m355_elemwise.h).
#include "m355_elemwise.h"

M355_DEFINE_LAUNCH(M355_ZI_NEGATIVEBINOMIAL, zi_negativebinomial)
