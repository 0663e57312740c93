"""Round 3: 1e9 draws of the discrete families with scalar parameters (k_rvs_inv) and of the same parameters passed as
(1024,) columns (per-lane samplers), int64 output."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
for name, p in (("poisson", (100.0,)), ("binomial", (1000.0, 0.3)), ("negativebinomial", (20.0, 0.05)), ("negativebinomial", (2.1, 0.375)),
                ("betabinomial", (100.0, 20.0, 20.0)), ("hypergeometric", (50.0, 20.0, 10.0)), ("zi_poisson", (0.3, 10.0)),
                ("zi_binomial", (0.5, 20.0, 0.5)), ("zi_negativebinomial", (0.7, 5.0, 0.4))):
    ms = t(lambda: _dispatch.rvs(name, p, size=n, random_state=3, as_torch=True))
    m = 100_000_000
    cols = tuple(torch.full((1, 1024), v, dtype=torch.float64, device="cuda") for v in p)
    try:
        ms_a = t(lambda: _dispatch.rvs(name, cols, size=(m // 1024, 1024), random_state=3, as_torch=True), reps=2) * (n / (m // 1024 * 1024))
    except Exception as e:  # noqa: BLE001
        ms_a = None
    print(json.dumps({"fn": f"{name}.rvs{p}", "n": n, "ms_scalar": ms, "draws_per_s": n / ms * 1e3, "ms_columns_per_1e9": ms_a}), flush=True)
