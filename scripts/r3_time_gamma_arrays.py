"""Round 3: gamma / beta / studentt rvs with array parameters (1e9 draws): k_rvs_mt_arr, k_rvs_bb_arr."""
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
a = torch.exp(torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 4.0)  # 1 .. 55
for tag, f in (("gamma, alpha (n,) log-uniform 1..55", lambda: _dispatch.rvs("gamma", (a, 1.0), size=n, random_state=3, as_torch=True)),
               ("gamma, alpha (n,) all 2.5", lambda: _dispatch.rvs("gamma", (torch.full_like(a, 2.5), 1.0), size=n, random_state=3, as_torch=True)),
               ("beta, a (n,) 1..55, b = 5", lambda: _dispatch.rvs("beta", (a, 5.0), size=n, random_state=3, as_torch=True)),
               ("beta, a (n,) all 2, b = 5", lambda: _dispatch.rvs("beta", (torch.full_like(a, 2.0), 5.0), size=n, random_state=3, as_torch=True)),
               ("studentt, nu (n,) 2..110", lambda: _dispatch.rvs("studentt", (2.0 * a, 0.0, 1.0), size=n, random_state=3, as_torch=True))):
    ms = t(f)
    print(json.dumps({"fn": tag, "n": n, "ms": ms, "draws_per_s": n / ms * 1e3}), flush=True)
