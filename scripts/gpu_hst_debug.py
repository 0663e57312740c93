import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
assert torch.cuda.is_available()
from distributions_amd import _lib, halfstudentt
from tests import harness as H
from oracle import restatement as R
L = _lib.lib()
rng = np.random.default_rng(33)
n = 40037
q = rng.uniform(0, 1, n)
q[:2000] = np.exp(rng.uniform(-20, 0, 2000)); q[2000:4000] = 1 - np.exp(rng.uniform(-20, 0, 2000))
for params in ((5.0, 2.0), (40.0, 1e-2)):
    for fn in ("ppf", "isf"):
        L.m355_set_tab_min_n(2**62); exact = getattr(halfstudentt, fn)(q, *params)
        L.m355_set_tab_min_n(1); tab = getattr(halfstudentt, fn)(q, *params)
        L.m355_set_tab_min_n(2**21)
        ok = H.ppf_close("halfstudentt", fn, params, tab, exact) | H.ppf_close("halfstudentt", fn, params, exact, tab)
        bad = np.where(~ok)[0]
        print(params, fn, "bad", bad.size)
        for i in bad[:8]:
            print("   q %.17g tab %.17g exact %.17g rel %.3e" % (q[i], tab[i], exact[i], abs(tab[i]-exact[i])/abs(exact[i])))
