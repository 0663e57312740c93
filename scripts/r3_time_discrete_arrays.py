"""Round 3: poisson / binomial rvs with array parameters (1e9 draws): k_rvs_compact_arr."""
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
mu100 = torch.full((n,), 100.0, dtype=torch.float64, device="cuda")
ms = t(lambda: _dispatch.rvs("poisson", (mu100,), size=n, random_state=3, as_torch=True))
print(json.dumps({"fn": "poisson.rvs, mu (n,) all 100", "n": n, "ms": ms, "draws_per_s": n / ms * 1e3}), flush=True)
mu = torch.exp(torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 6.0 + 2.5)  # 12 .. 4900
ms = t(lambda: _dispatch.rvs("poisson", (mu,), size=n, random_state=3, as_torch=True))
print(json.dumps({"fn": "poisson.rvs, mu (n,) log-uniform 12..4900", "n": n, "ms": ms, "draws_per_s": n / ms * 1e3}), flush=True)
mu = torch.exp(torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 6.0 - 1.0)  # 0.37 .. 150: mixes inversion and PTRS
ms = t(lambda: _dispatch.rvs("poisson", (mu,), size=n, random_state=3, as_torch=True))
print(json.dumps({"fn": "poisson.rvs, mu (n,) log-uniform 0.37..150", "n": n, "ms": ms, "draws_per_s": n / ms * 1e3}), flush=True)
del mu, mu100
pr = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 0.8 + 0.1
ms = t(lambda: _dispatch.rvs("binomial", (1000.0, pr), size=n, random_state=3, as_torch=True))
print(json.dumps({"fn": "binomial.rvs, n = 1000, p (n,) uniform 0.1..0.9", "n": n, "ms": ms, "draws_per_s": n / ms * 1e3}), flush=True)
