"""binomial.logcdf at n = 1000, p ~ 0.5: GPU vs SciPy oracle vs mpmath (prints; not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpmath as mp
from oracle import restatement as R
from distributions_amd import binomial
mp.mp.dps = 40
p = np.linspace(0.05, 0.95, 1024)
rng = np.random.default_rng(0)
cols = rng.integers(400, 620, 4000)
x = rng.binomial(1000, p[cols]).astype(np.float64)
got = binomial.logcdf(x, 1000.0, p[cols])
exp = R.binomial_logcdf(x, 1000.0, p[cols])
bad = np.argsort(-np.abs(got - exp))[:6]
for i in bad:
    k, pp = int(x[i]), float(p[cols[i]])
    true = mp.log(mp.betainc(1000 - k, k + 1, 0, 1 - mp.mpf(pp), regularized=True))
    print(k, pp, "gpu", got[i], "scipy", exp[i], "mp", float(true), "gpu-mp", float(got[i] - true), "scipy-mp", float(exp[i] - true))
