"""Round 3: pure-write ceilings next to the output-only rvs kernels (8 GB each)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributions_amd import _dispatch

n = 1_000_000_000
buf = torch.empty(n, dtype=torch.float64, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: buf.fill_(1.5))
print(json.dumps({"what": "torch fill_ 8 GB", "ms": ms, "TB_per_s": 8e-3 * n / 1e9 / ms * 1e3 / 1e3}), flush=True)
ms = t(lambda: buf.zero_())
print(json.dumps({"what": "torch zero_ 8 GB", "ms": ms, "TB_per_s": 8 * n / ms / 1e9}), flush=True)
for name, p in (("uniform", (0.0, 1.0)), ("exponential", (1.0,)), ("normal", (0.0, 1.0)), ("cauchy", (0.0, 1.0))):
    ms = t(lambda: _dispatch.rvs(name, p, size=n, random_state=3, as_torch=True, out=buf) if False else _dispatch.rvs(name, p, size=n, random_state=3, as_torch=True), reps=3)
    print(json.dumps({"what": f"{name}.rvs 1e9", "ms": ms, "TB_per_s": 8 * n / ms / 1e9}), flush=True)
