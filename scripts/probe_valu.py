"""Where do the VALU-bound kernels spend time?  Times <dist>.<fn> on constant-x arrays (no
divergence) across the argument range.  Usage: python scripts/probe_valu.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributions_amd import raw
from bench_configs import timed

n = 20_000_000
dev = torch.device("cuda", 0)
out = torch.empty(n, dtype=torch.float64, device=dev)
def run(name, fn, p, vals):
    row = []
    for v in vals:
        x = torch.full((n,), v, dtype=torch.float64, device=dev)
        ms = timed(torch, lambda: raw.elemwise_into(name, fn, x, p, out), reps=3)
        row.append(f"{v:g}:{ms:.2f}")
    print(name, fn, " ".join(row), flush=True)
run("studentt", "logcdf", (5.0, 0.0, 2.0), [-20, -6, -4, -3, -2, -1, -0.3, 0.0, 0.3, 1, 2, 3, 4, 6, 20])
run("beta", "logcdf", (2.0, 5.0), [0.01, 0.1, 0.2, 0.3, 0.4, 0.5, 0.7, 0.9, 0.99])
run("gamma", "logcdf", (2.5, 1.3), [0.01, 0.1, 0.5, 1, 1.5, 2, 3, 5, 10])
qs = [1e-6, 0.01, 0.1, 0.3, 0.5, 0.7, 0.9, 0.99, 1 - 1e-6]
run("gamma", "ppf", (2.5, 1.3), qs)
run("beta", "ppf", (2.0, 5.0), qs)
run("studentt", "ppf", (5.0, 0.0, 2.0), qs)
run("normal", "ppf", (0.3, 1.7), qs)
