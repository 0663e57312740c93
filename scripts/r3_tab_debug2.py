"""Round-3: which quantile tables of config 6's families really serve (prints; not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distributions_amd import _lib
import importlib
import tests.test_gpu_tables_full as T

L = _lib.lib()
rng = np.random.default_rng(1)
q = rng.uniform(0, 1, 3_000_000)
for name, params in (("betascaled", (2.0, 5.0, -1.0, 3.0)), ("inversegamma", (5.0, 2.0)), ("scaledinversechisquared", (5.0, 2.0)),
                     ("beta", (2.0, 5.0)), ("chisquared", (5.0,)), ("halfstudentt", (5.0, 2.0))):
    m = importlib.import_module(f"distributions_amd.{name}")
    exact = T._exact(L, m.ppf, q, *params)
    tab = m.ppf(q, *params)
    w = np.log(q / (1 - q))
    differs = tab != exact
    bins = np.linspace(-21, 21, 22)
    frac = [round(float(np.mean(differs[(w >= a) & (w < b)])), 2) if np.any((w >= a) & (w < b)) else None for a, b in zip(bins[:-1], bins[1:])]
    print(name, "ppf: frac(tab != exact) overall", round(float(differs.mean()), 3), "by logit(q) bin", frac, flush=True)
