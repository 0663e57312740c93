import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from distributions_amd import _lib, gamma, beta
from oracle import restatement as R
L = _lib.lib()
dev = torch.device("cuda:0")
n = 100_000_000
def t(f):
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    f(); e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 5
for a in (2.5, 0.9, 0.3, 0.05, 0.01):
    x = torch.distributions.Gamma(torch.tensor(a, dtype=torch.float64, device=dev), torch.tensor(1.0, dtype=torch.float64, device=dev)).sample((n,)).clamp_(min=1e-300)
    for fn in ("logcdf", "logsf"):
        f = getattr(gamma, fn)
        L.m355_set_tab_min_n(2**62); exact = f(x, a, 1.0); te = t(lambda: f(x, a, 1.0))
        L.m355_set_tab_min_n(2**21); tab = f(x, a, 1.0); tt = t(lambda: f(x, a, 1.0))
        same = (tab == exact)
        d = ((tab - exact).abs() / exact.abs().clamp(min=1)).max()
        print("gamma", a, fn, "identical %.4f maxdiff %.2e  exact %.3f ms tab %.3f ms" % (float(same.double().mean()), float(d), te, tt))
for a, b in ((2.0, 5.0), (0.3, 4.0), (3.0, 0.2), (0.5, 0.5), (0.05, 0.02)):
    x = torch.distributions.Beta(torch.tensor(a, dtype=torch.float64, device=dev), torch.tensor(b, dtype=torch.float64, device=dev)).sample((n,))
    for fn in ("logcdf", "logsf"):
        f = getattr(beta, fn)
        L.m355_set_tab_min_n(2**62); exact = f(x, a, b); te = t(lambda: f(x, a, b))
        L.m355_set_tab_min_n(2**21); tab = f(x, a, b); tt = t(lambda: f(x, a, b))
        same = (tab == exact)
        fin = torch.isfinite(exact)
        d = ((tab - exact).abs() / exact.abs().clamp(min=1))[fin].max()
        print("beta", a, b, fn, "identical %.4f maxdiff %.2e  exact %.3f ms tab %.3f ms" % (float(same.double().mean()), float(d), te, tt))
