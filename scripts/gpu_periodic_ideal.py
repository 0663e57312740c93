"""What the row-periodic special-function kernels could cost at best: the scalar-parameter (k_binned) time of the same x
at 16 parameter points across the (1024,) rows of bench_configs.periodic_special, averaged."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import make_inputs  # noqa: E402
from bench_configs import timed  # noqa: E402
from distributions_amd import raw  # noqa: E402

dev = torch.device("cuda", 0)
n = 20_000_000
xs = make_inputs(torch, n, dev)
q = torch.empty(n, dtype=torch.float64, device=dev)
raw.rvs_into("uniform", (0.0, 1.0), q, seed=5)
q.clamp_(1e-12, 1 - 1e-12)
out = torch.empty(n, dtype=torch.float64, device=dev)
lin = lambda lo, hi, k: [lo + (hi - lo) * (j + 0.5) / k for j in range(k)]
K = 16
cases = {"gamma": list(zip(lin(0.5, 9.0, K), lin(0.3, 3.0, K))), "beta": list(zip(lin(0.5, 9.0, K), lin(1.0, 10.0, K))),
         "studentt": [(nu, 0.0, 2.0) for nu in lin(2.0, 30.0, K)]}
for name, plist in cases.items():
    for fn in ("logcdf", "ppf"):
        src = q if fn == "ppf" else xs[name]
        ms = [timed(torch, lambda: raw.elemwise_into(name, fn, src, p, out), reps=3) for p in plist]
        print(json.dumps({"fn": f"{name}.{fn}", "n": n, "scalar_ms_per_point": [round(m, 3) for m in ms],
                          "ideal_ms_per_1e8": sum(ms) / len(ms) * 1e8 / n}), flush=True)
