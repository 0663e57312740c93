"""Round 3: compound count families' rvs with array parameters (1e9 draws): k_rvs_zi_arr for zi_poisson / zi_binomial,
the per-lane loop of k_rvs for the families with a gamma stage."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributions_amd import _dispatch

def t(f, reps=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

n = 1_000_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
u = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
p = 0.02 + 0.1 * u          # negative-binomial mean n (1 - p) / p from 160 to 980 at n = 20
mu = torch.exp(2.5 + 4.0 * u)  # 12 .. 660
cases = (
    ("negativebinomial.rvs, n = 20, p (n,) 0.02..0.12", "negativebinomial", (20.0, p)),
    ("negativebinomial.rvs, n = 2.1, p (n,) all 0.375 (small means: draws finished on the spot)", "negativebinomial", (2.1, torch.full_like(u, 0.375))),
    ("zi_poisson.rvs, psi = 0.3, mu (n,) 12..660", "zi_poisson", (0.3, mu)),
    ("zi_negativebinomial.rvs, psi = 0.7, n = 20, p (n,) 0.02..0.12", "zi_negativebinomial", (0.7, 20.0, p)),
    ("zi_binomial.rvs, psi = 0.5, n = 1000, p (n,) 0.1..0.9", "zi_binomial", (0.5, 1000.0, 0.1 + 0.8 * u)),
    ("betabinomial.rvs, n = 1000, alpha (n,) 2..22, beta = 20", "betabinomial", (1000.0, 2.0 + 20.0 * u, 20.0)),
)
for tag, name, params in cases:
    ms = t(lambda: _dispatch.rvs(name, params, size=n, random_state=3, as_torch=True))
    print(json.dumps({"fn": tag, "n": n, "ms": ms, "draws_per_s": n / ms * 1e3}), flush=True)
