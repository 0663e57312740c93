"""Round 3: gamma.rvs against the shape (1e9 draws, scalar parameters)."""
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
for a in (1.0, 1.5, 2.5, 5.0, 10.0, 100.0, 1e4):
    ms = t(lambda: _dispatch.rvs("gamma", (a, 1.0), size=n, random_state=3, as_torch=True))
    print(json.dumps({"fn": f"gamma.rvs({a}, 1)", "n": n, "ms": ms, "draws_per_s": n / ms * 1e3}), flush=True)
for nu in (5.0, 30.0):
    ms = t(lambda: _dispatch.rvs("studentt", (nu, 0.0, 1.0), size=n, random_state=3, as_torch=True))
    print(json.dumps({"fn": f"studentt.rvs({nu}, 0, 1)", "n": n, "ms": ms, "draws_per_s": n / ms * 1e3}), flush=True)
