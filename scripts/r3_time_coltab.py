"""Times gamma / beta / studentt logcdf and ppf with (1024,) row-periodic parameters at N = 1e8 (HIP events, whole call)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_configs import timed
from bench import make_inputs
from distributions_amd import raw
dev = torch.device("cuda", 0)
n = (100_000_000 // 1024) * 1024
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["gamma", "beta", "studentt"]
xs = make_inputs(torch, n, dev)
q = torch.empty(n, dtype=torch.float64, device=dev); raw.rvs_into("uniform", (0.0, 1.0), q, seed=5); q.clamp_(1e-12, 1 - 1e-12)
out = torch.empty(n, dtype=torch.float64, device=dev)
ews = raw.elemwise_workspace(dev)
lin = lambda lo, hi: torch.linspace(lo, hi, 1024, dtype=torch.float64, device=dev)
cases = {"gamma": (lin(0.5, 9.0), lin(0.3, 3.0)), "beta": (lin(0.5, 9.0), lin(1.0, 10.0)), "studentt": (lin(2.0, 30.0), 0.0, 2.0)}
for name in names:
    for fn in ("logcdf", "ppf"):
        src = q if fn == "ppf" else xs[name]
        ms = timed(torch, lambda: raw.elemwise_into(name, fn, src, cases[name], out, ews), reps=5)
        print(json.dumps({"fn": f"{name}.{fn}", "ms": round(ms, 4)}), flush=True)
