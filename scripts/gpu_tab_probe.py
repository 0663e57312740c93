"""k_tab against the exact kernels: max scaled difference and timings (scalar parameters)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from distributions_amd import _lib, beta, gamma, studentt

L = _lib.lib()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
g = torch.Generator(device="cpu").manual_seed(5)


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


cases = [
    ("gamma", gamma, (2.5, 1.3), lambda: torch.distributions.Gamma(2.5, 1.3).sample((n,)).double()),
    ("gamma", gamma, (0.3, 1.0), lambda: torch.distributions.Gamma(0.3, 1.0).sample((n,)).double()),
    ("gamma", gamma, (400.0, 2.0), lambda: torch.distributions.Gamma(400.0, 2.0).sample((n,)).double()),
    ("beta", beta, (2.0, 5.0), lambda: torch.distributions.Beta(2.0, 5.0).sample((n,)).double()),
    ("beta", beta, (0.5, 0.5), lambda: torch.distributions.Beta(0.5, 0.5).sample((n,)).double()),
    ("beta", beta, (30.0, 0.7), lambda: torch.distributions.Beta(30.0, 0.7).sample((n,)).double()),
    ("studentt", studentt, (5.0, 0.0, 2.0), lambda: (torch.distributions.StudentT(5.0).sample((n,)) * 2.0).double()),
    ("studentt", studentt, (1.0, 1.0, 0.5), lambda: (1.0 + torch.distributions.StudentT(1.0).sample((n,)) * 0.5).double()),
]
for name, mod, params, draw in cases:
    x = draw().to(dev)
    for fn in ("logcdf", "logsf", "cdf", "sf"):
        f = getattr(mod, fn)
        L.m355_set_tab_min_n(2**62)
        exact = f(x, *params)
        t_exact = timed(lambda: f(x, *params), reps=2)
        L.m355_set_tab_min_n(-1)
        tab = f(x, *params)
        t_tab = timed(lambda: f(x, *params))
        d = (tab - exact).abs() / exact.abs().clamp(min=1.0)
        d = torch.where(torch.isnan(d), torch.zeros_like(d), d)
        same_nan = bool((torch.isnan(tab) == torch.isnan(exact)).all())
        print(json.dumps({"dist": name, "params": params, "fn": fn, "n": n, "ms_exact": round(t_exact, 3), "ms_tab": round(t_tab, 3),
                          "max_scaled_diff": float(d.max()), "identical_frac": float((tab == exact).double().mean()), "nan_match": same_nan}), flush=True)
    del x
