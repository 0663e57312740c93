import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from distributions_amd import _lib, gamma
L = _lib.lib()
dev = torch.device("cuda:0")
n = 100_000_000
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
x = torch.distributions.Gamma(2.5, 1.3).sample((n,)).double().to(dev)
xm = torch.full((n,), 1.7, dtype=torch.float64, device=dev)
xs, _ = torch.sort(x[:n])
out = torch.empty_like(x)
from distributions_amd import raw
for name, xx in (("random", x), ("constant", xm), ("sorted", xs)):
    for fn in ("logcdf", "cdf"):
        f = getattr(gamma, fn)
        print(name, fn, round(timed(lambda: f(xx, 2.5, 1.3)), 3), "ms;  n=4e6:", round(timed(lambda: f(xx[:4_000_000], 2.5, 1.3)), 4))
