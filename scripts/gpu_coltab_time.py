import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from distributions_amd import _lib, gamma, beta, studentt
L = _lib.lib()
dev = torch.device("cuda:0")
P, R = 1024, 97656
g = torch.Generator(device=dev); g.manual_seed(3)
def t(f):
    f(); torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 5
al = torch.empty(P, dtype=torch.float64, device=dev).uniform_(1.0, 12.0, generator=g)
be = torch.empty(P, dtype=torch.float64, device=dev).uniform_(0.5, 3.0, generator=g)
cases = []
xg = torch.distributions.Gamma(al, be).sample((R,))
cases.append(("gamma", gamma, xg, (al, be)))
a2 = torch.empty(P, dtype=torch.float64, device=dev).uniform_(1.0, 9.0, generator=g)
b2 = torch.empty(P, dtype=torch.float64, device=dev).uniform_(1.0, 9.0, generator=g)
xb = torch.distributions.Beta(a2, b2).sample((R,))
cases.append(("beta", beta, xb, (a2, b2)))
nu = torch.empty(P, dtype=torch.float64, device=dev).uniform_(2.0, 40.0, generator=g)
mu = torch.zeros(P, dtype=torch.float64, device=dev); sg = torch.ones(P, dtype=torch.float64, device=dev) * 1.5
xs = torch.distributions.StudentT(nu, mu, sg).sample((R,))
cases.append(("studentt", studentt, xs, (nu, mu, sg)))
q = torch.rand(R, P, dtype=torch.float64, device=dev, generator=g)
for name, m, x, params in cases:
    for fn in ("logcdf", "logsf", "ppf"):
        f = getattr(m, fn); arg = q if fn == "ppf" else x
        L.m355_set_tab_min_n(2**59); exact = f(arg, *params); te = t(lambda: f(arg, *params))
        L.m355_set_tab_min_n(2**21); tab = f(arg, *params); tt = t(lambda: f(arg, *params))
        fin = torch.isfinite(exact)
        d = ((tab - exact).abs() / exact.abs().clamp(min=1))[fin].max()
        print(name, fn, "identical %.4f maxdiff %.2e  exact %.3f ms  coltab %.3f ms" % (float((tab == exact).double().mean()), float(d), te, tt), flush=True)
