"""A/B timing of gamma.logcdf / ppf and poisson.logcdf at large shape parameters: the current library against a build
of the previous commit placed at tools/_old_lib/libm355dist.so (not committed).  usage: python scripts/gpu_temme_ab.py [old]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

import distributions_amd._lib as L

if len(sys.argv) > 1 and sys.argv[1] == "old":
    L.LIB_PATH = "tools/_old_lib/libm355dist.so"
from distributions_amd import gamma, poisson  # noqa: E402

dev = torch.device("cuda:0")
n = 10_000_000
g = torch.Generator(device="cpu").manual_seed(3)
z = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
u = torch.rand(n, generator=g, dtype=torch.float64).to(dev).clamp_(1e-9, 1 - 1e-9)


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


for a in (30.0, 100.0, 1e3, 1e4, 1e6, 1e8):
    x = (a + a**0.5 * z).clamp_(min=a * 0.5)
    row = {"lib": "old" if len(sys.argv) > 1 else "new", "a": a, "n": n}
    row["gamma.logcdf_ms"] = timed(lambda: gamma.logcdf(x, a, 1.0))
    row["gamma.ppf_ms"] = timed(lambda: gamma.ppf(u, a, 1.0), reps=2)
    row["poisson.logcdf_ms"] = timed(lambda: poisson.logcdf(torch.floor(x), a))
    print(json.dumps(row), flush=True)
