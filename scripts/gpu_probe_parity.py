"""Lists every golden-vector point where the HIP path differs from the committed expectation by more than the strict
bar |a-b| <= 1e-12 max(1,|b|) (no family masks): the input for the per-family whitelists in tests/harness.py."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import harness as H  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "elemwise.npz"))
cases = sorted({k.split("__")[0] for k in g.files})
ALL = ["logpdf", "logcdf", "logsf", "pdf", "cdf", "sf", "ppf", "isf"]
for key in cases:
    name = key.split("_", 1)[1]
    params = list(g[f"{key}__params"])
    m = importlib.import_module(f"distributions_amd.{name}")
    for fn in ALL:
        pts = g[f"{key}__q"] if fn in ("ppf", "isf") else g[f"{key}__x"]
        exp = g[f"{key}__{fn}"]
        got = getattr(m, fn)(pts, *params)
        ok = H.close(got, exp, 1e-12) | (np.isnan(exp) & (fn in ("ppf", "isf")))
        if not ok.all():
            with np.errstate(all="ignore"):
                err = np.abs(got - exp) / np.maximum(1, np.abs(exp))
            bad = ~ok
            order = np.argsort(-np.nan_to_num(err[bad], nan=np.inf))[:3]
            print(key, fn, [float(p) for p in params], "nbad", int(bad.sum()), "pts", pts[bad][order], "got", got[bad][order],
                  "exp", exp[bad][order], "err", err[bad][order], flush=True)
print("probe done")
