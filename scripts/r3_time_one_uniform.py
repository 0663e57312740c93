"""Round 3: the one-uniform / one-normal inverse-cdf samplers (k_rvs_pairs), 1e9 draws each."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributions_amd import _dispatch

n = 1_000_000_000
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, p in (("uniform", (0.0, 1.0)), ("exponential", (1.0,)), ("laplace", (0.0, 1.0)), ("logistic", (0.0, 1.0)),
                ("gumbel", (0.0, 1.0)), ("weibull", (1.5, 2.0)), ("pareto", (3.0, 1.0)), ("cauchy", (0.0, 1.0)),
                ("lognormal", (0.0, 1.0)), ("halfnormal", (1.0,)), ("moyal", (0.0, 1.0)), ("geometric", (0.2,))):
    ms = t(lambda: _dispatch.rvs(name, p, size=n, random_state=3, as_torch=True))
    print(json.dumps({"fn": f"{name}.rvs{p}", "n": n, "ms": ms, "draws_per_s": n / ms * 1e3}), flush=True)
