"""k_tab quantiles against the exact kernels and SciPy."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from scipy import stats
from distributions_amd import _lib, beta, gamma, studentt
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
cases = [("studentt", studentt, (5.0, 0.0, 2.0), stats.t(5.0, 0.0, 2.0)), ("studentt", studentt, (1.0, 1.0, 0.5), stats.t(1.0, 1.0, 0.5)),
         ("studentt", studentt, (300.0, -2.0, 1.0), stats.t(300.0, -2.0, 1.0)),
         ("gamma", gamma, (2.5, 1.3), stats.gamma(2.5, scale=1 / 1.3)), ("gamma", gamma, (1.0, 1.0), stats.gamma(1.0)),
         ("gamma", gamma, (400.0, 2.0), stats.gamma(400.0, scale=0.5)), ("beta", beta, (2.0, 5.0), stats.beta(2.0, 5.0)),
         ("beta", beta, (60.0, 1.5), stats.beta(60.0, 1.5)), ("beta", beta, (150.0, 300.0), stats.beta(150.0, 300.0))]
xs_dev = {}
for name, mod, params, sd in cases:
    if name == 'studentt':
        xs_dev[name + str(params)] = (params[1] + params[2] * torch.distributions.StudentT(params[0]).sample((n,))).double().to(dev)
    for fn in ("ppf", "isf", "logcdf", "logsf", "cdf"):
        f = getattr(mod, fn)
        if fn not in ("ppf", "isf") and name != "studentt":
            continue
        arg = q if fn in ("ppf", "isf") else xs_dev[name + str(params)]
        L.m355_set_tab_min_n(2**62); exact = f(arg, *params); t_exact = timed(lambda: f(arg, *params), reps=1)
        L.m355_set_tab_min_n(-1); tab = f(arg, *params); t_tab = timed(lambda: f(arg, *params))
        d = (tab - exact).abs() / exact.abs().clamp(min=1.0)
        d = torch.where(torch.isnan(d), torch.zeros_like(d), d)
        i = int(torch.argmax(d))
        qs = arg[:200000].cpu().numpy()
        ref = getattr(sd, fn)(qs)
        dr = np.abs(tab[:200000].cpu().numpy() - ref) / np.maximum(1.0, np.abs(ref))
        print(json.dumps({"dist": name, "params": params, "fn": fn, "ms_exact": round(t_exact, 3), "ms_tab": round(t_tab, 3), "max_rel_vs_exact": float(d.max()),
                          "at": float(arg[i]), "identical": float((tab == exact).double().mean()), "max_rel_vs_scipy": float(np.nanmax(dr))}), flush=True)
