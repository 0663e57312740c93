import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from distributions_amd import _lib, halfstudentt
L = _lib.lib()
dev = torch.device("cuda:0")
n = 100_000_000
def t(f):
    f(); torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 5
for nu, sg in ((5.0, 2.0), (1.0, 1.0), (30.0, 0.5)):
    x = (torch.distributions.StudentT(torch.tensor(nu, dtype=torch.float64, device=dev)).sample((n,)).abs() * sg)
    q = torch.rand(n, dtype=torch.float64, device=dev)
    for fn in ("cdf", "sf", "ppf", "logcdf", "logsf"):
        f = getattr(halfstudentt, fn); arg = q if fn == "ppf" else x
        L.m355_set_tab_min_n(2**59); exact = f(arg, nu, sg); te = t(lambda: f(arg, nu, sg))
        L.m355_set_tab_min_n(2**21); tab = f(arg, nu, sg); tt = t(lambda: f(arg, nu, sg))
        fin = torch.isfinite(exact)
        d = ((tab - exact).abs() / exact.abs().clamp(min=1))[fin].max()
        print("halfstudentt", nu, sg, fn, "identical %.4f maxdiff %.2e  exact %.3f ms  tab %.3f ms" % (float((tab == exact).double().mean()), float(d), te, tt), flush=True)
