import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy import stats
from distributions_amd import _dispatch
n = 100_000_000
for name, p, d in (("studentt", (5.0, 0.0, 2.0), stats.t(5, 0, 2)), ("studentt", (4.0, 0.3, 1.7), stats.t(4.0, 0.3, 1.7))):
    x = _dispatch.rvs(name, p, size=n, random_state=3, as_torch=True)
    print(name, p, "finite", bool(torch.isfinite(x).all()), "min", float(x.min()), "max", float(x.max()))
    z = (x - p[1]) / p[2]
    for lim in (1e-4, 5.7e-4, 1e-2):
        got = int((z.abs() < lim).sum()); exp = (d.cdf(p[1] + lim * p[2]) - d.cdf(p[1] - lim * p[2])) * n
        print("  |z| <", lim, got, "expected", round(exp))
    for q in (1e-9, 1e-7, 1e-5, 1e-3):
        lo, hi = d.ppf(q), d.isf(q)
        print("  tails at", q, int((x < lo).sum()), int((x > hi).sum()), "expected", q * n)
    print("  exact zeros", int((x == p[1]).sum()), "duplicates among first 1e6:", 1_000_000 - len(np.unique(x[:1_000_000].cpu().numpy())))
