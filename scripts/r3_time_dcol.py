"""Time cdf / logcdf of discrete families with (1024,) parameters at N = 97656 x 1024 (k_dcol vs the exact kernels)."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from distributions_amd import _lib, raw  # noqa: E402

dev = torch.device("cuda", 0)
L = _lib.lib()
P, R = 1024, 97656
rng = np.random.default_rng(3)
cases = {
    "poisson": (np.linspace(5.0, 200.0, P),),
    "binomial": (np.full(P, 1000.0), np.linspace(0.05, 0.95, P)),
    "negativebinomial": (np.linspace(2.0, 20.0, P), np.full(P, 0.1)),
}
ws = raw.elemwise_workspace(dev)
for name, pp in cases.items():
    pd = tuple(torch.from_numpy(a).to(dev) for a in pp)
    x = torch.empty(R * P, dtype=torch.int64, device=dev)
    raw.rvs_into(name, pd, x.view(R, P).reshape(-1), seed=5) if False else None
    # draws with per-column parameters through the eager API (int64)
    import importlib
    m = importlib.import_module(f"distributions_amd.{name}")
    x = m.rvs(*[p.reshape(1, P) for p in pd], size=(R, P), random_state=5).reshape(-1).contiguous()
    out = torch.empty(R * P, dtype=torch.float64, device=dev)
    qd = torch.rand(R * P, dtype=torch.float64, device=dev)
    for fn in ("logcdf", "cdf", "ppf"):
        res = {}
        for label, thr in (("exact", 2**62), ("k_dcol", -1)):
            old = L.m355_set_tab_min_n(thr)
            try:
                arg = qd if fn == "ppf" else x
                reps = 2 if (fn == "ppf" and label == "exact") else 5
                for _ in range(1 if reps == 2 else 2):
                    raw.elemwise_into(name, fn, arg, pd, out, workspace=ws)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                ev[0].record()
                for _ in range(reps):
                    raw.elemwise_into(name, fn, arg, pd, out, workspace=ws)
                ev[1].record()
                torch.cuda.synchronize()
                res[label] = ev[0].elapsed_time(ev[1]) / reps
                res[label + "_sum"] = float(out.sum().item())
            finally:
                L.m355_set_tab_min_n(old)
        print(json.dumps({"dist": name, "fn": fn, "n": R * P, "layout": "params (1024,), int64 observations", "ms_exact": res["exact"],
                          "ms_k_dcol": res["k_dcol"], "identical_sums": res["exact_sum"] == res["k_dcol_sum"]}), flush=True)
