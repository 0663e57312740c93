"""Time the fused sum_logcdf / sum_logsf launches (N = 1e8, scalar parameters) with HIP events; one JSON line each."""
import json
import sys

import torch

sys.path.insert(0, ".")
from bench import FAMILIES, make_inputs  # noqa: E402
from distributions_amd import raw  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
xs = make_inputs(torch, n, dev)
ws = raw.sumlogp_workspace(dev)
slot = torch.zeros(1, dtype=torch.float64, device=dev)
for name, p, *_ in FAMILIES:
    for fn in ("logpdf", "logcdf", "logsf"):
        for _ in range(3):
            raw.sumlogp_into(name, fn, xs[name], p, slot, ws)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10):
            raw.sumlogp_into(name, fn, xs[name], p, slot, ws)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 10
        print(json.dumps({"dist": name, "fn": f"sum_{fn}", "n": n, "ms": ms, "GB/s": 8e-6 * n / ms, "total": float(slot.item())}), flush=True)
