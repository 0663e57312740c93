"""Round 3: normal.rvs with array parameters (1e9 draws): mu of shape (n,), mu of shape (1024,) over rows, scalars."""
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

R, P = 976_562, 1024
n = R * P
mu_n = torch.randn(n, dtype=torch.float64, device="cuda")
mu_p = torch.randn(1, P, dtype=torch.float64, device="cuda")
sg_p = torch.rand(1, P, dtype=torch.float64, device="cuda") + 0.5
for tag, f in (("scalars", lambda: _dispatch.rvs("normal", (0.0, 1.0), size=n, random_state=3, as_torch=True)),
               ("mu (n,)", lambda: _dispatch.rvs("normal", (mu_n, 1.0), size=n, random_state=3, as_torch=True)),
               ("mu, sigma (1024,) columns", lambda: _dispatch.rvs("normal", (mu_p, sg_p), size=(R, P), random_state=3, as_torch=True)),
               ("mu (n,), odd offset: per-lane kernel", lambda: _dispatch.rvs("normal", (mu_n, 1.0), size=n, random_state=(3, 1), as_torch=True))):
    ms = t(f)
    print(json.dumps({"fn": f"normal.rvs, {tag}", "n": n, "ms": ms, "draws_per_s": n / ms * 1e3}), flush=True)
