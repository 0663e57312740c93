import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
assert torch.cuda.is_available()
from distributions_amd import _lib, beta
from oracle import restatement as R
L = _lib.lib()
np.set_printoptions(precision=12, linewidth=200)
xs = np.concatenate([[8.18869400e-17, 4.38890135e-10, 5.03003580e-10], np.logspace(-16, -3, 27)])
for fn in ("logsf", "sf"):
    L.m355_set_tab_min_n(2**62); exact = getattr(beta, fn)(xs, 0.5, 0.5)
    L.m355_set_tab_min_n(1); tab = getattr(beta, fn)(xs, 0.5, 0.5)
    with np.errstate(all="ignore"):
        exp = R.TABLE["beta_" + fn](xs, 0.5, 0.5)
    for i in range(xs.size):
        print(fn, "%.6e exact % .12e tab-exact % .3e oracle-exact % .3e" % (xs[i], exact[i], tab[i] - exact[i], exp[i] - exact[i]))
