"""Divergence check: the config-2 random inputs, shuffled (as benchmarked) vs sorted (wave-homogeneous)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributions_amd import raw
from bench_configs import timed

n = 20_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(4)
out = torch.empty(n, dtype=torch.float64, device=dev)
z = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
c = sum(torch.randn(n, dtype=torch.float64, device=dev, generator=g) ** 2 for _ in range(5))
xt = 2.0 * z / torch.sqrt(c / 5.0)
u = torch.rand(n, dtype=torch.float64, device=dev, generator=g)
xg = -(torch.log(u) + torch.log(torch.rand(n, dtype=torch.float64, device=dev, generator=g))) / 1.3
q = torch.rand(n, dtype=torch.float64, device=dev, generator=g).clamp_(1e-12, 1 - 1e-12)
for name, fn, p, x in (("studentt", "logcdf", (5.0, 0.0, 2.0), xt), ("gamma", "logcdf", (2.5, 1.3), xg),
                       ("beta", "logcdf", (2.0, 5.0), u), ("gamma", "ppf", (2.5, 1.3), q), ("beta", "ppf", (2.0, 5.0), q),
                       ("studentt", "ppf", (5.0, 0.0, 2.0), q)):
    xs = torch.sort(x).values
    a = timed(torch, lambda: raw.elemwise_into(name, fn, x, p, out), reps=3)
    b = timed(torch, lambda: raw.elemwise_into(name, fn, xs, p, out), reps=3)
    print(f"{name}.{fn}: shuffled {a:.2f} ms  sorted {b:.2f} ms  (n={n})", flush=True)
