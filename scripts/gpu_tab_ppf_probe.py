"""k_tab quantiles against the exact kernels and SciPy."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from scipy import stats
from distributions_amd import _lib, beta, gamma
L = _lib.lib()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
g = torch.Generator(device="cpu").manual_seed(5)
q = torch.rand(n, generator=g, dtype=torch.float64).to(dev)
def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
cases = [("gamma", gamma, (2.5, 1.3), stats.gamma(2.5, scale=1 / 1.3)), ("gamma", gamma, (1.0, 1.0), stats.gamma(1.0)),
         ("gamma", gamma, (400.0, 2.0), stats.gamma(400.0, scale=0.5)), ("beta", beta, (2.0, 5.0), stats.beta(2.0, 5.0)),
         ("beta", beta, (60.0, 1.5), stats.beta(60.0, 1.5)), ("beta", beta, (150.0, 300.0), stats.beta(150.0, 300.0))]
for name, mod, params, sd in cases:
    for fn in ("ppf", "isf"):
        f = getattr(mod, fn)
        L.m355_set_tab_min_n(2**62); exact = f(q, *params); t_exact = timed(lambda: f(q, *params), reps=1)
        L.m355_set_tab_min_n(-1); tab = f(q, *params); t_tab = timed(lambda: f(q, *params))
        d = (tab - exact).abs() / exact.abs()
        d = torch.where(torch.isnan(d), torch.zeros_like(d), d)
        i = int(torch.argmax(d))
        qs = q[:200000].cpu().numpy()
        ref = getattr(sd, fn)(qs)
        dr = np.abs(tab[:200000].cpu().numpy() - ref) / np.abs(ref)
        print(json.dumps({"dist": name, "params": params, "fn": fn, "ms_exact": round(t_exact, 3), "ms_tab": round(t_tab, 3), "max_rel_vs_exact": float(d.max()),
                          "at_q": float(q[i]), "identical": float((tab == exact).double().mean()), "max_rel_vs_scipy": float(np.nanmax(dr))}), flush=True)
