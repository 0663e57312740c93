"""Generates tests/golden/elemwise.npz from the CPU oracle (oracle/restatement.py), which is itself
pinned against the reference's SciPy test grids by tests/test_oracle_pinning.py.  The reference
cannot be imported here (PyTensor absent), so these vectors are the oracle's outputs
on the reference's own evaluation points (tests/helper_scipy.py:75-84 `extended_vals`, the isf
points of :152) plus a seeded random block per case.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import restatement as R  # noqa: E402
from tests import harness as H  # noqa: E402

ALL_FNS = ["logpdf", "logcdf", "logsf", "pdf", "cdf", "sf", "ppf", "isf"]
FNS = {g[0]: ALL_FNS for g in H.grids()}


def main():
    out = {}
    rng = np.random.default_rng(20260313)
    for ci, (name, sp_dist, params, sp_params, support, _kw) in enumerate(H.grids()):
        d = sp_dist(**sp_params)
        xs = np.concatenate([H.extended_vals(d, support),
                             np.asarray(d.rvs(256, random_state=rng), dtype=np.float64)])
        qs = np.concatenate([H.ISF_POINTS, rng.uniform(0, 1, 128), [1e-12, 1 - 1e-12, 1e-300]])
        key = f"c{ci:02d}_{name}"
        out[f"{key}__params"] = np.asarray(params, dtype=np.float64)
        out[f"{key}__x"] = xs
        out[f"{key}__q"] = qs
        for fn in FNS[name]:
            pts = qs if fn in ("ppf", "isf") else xs
            with np.errstate(all="ignore"):
                out[f"{key}__{fn}"] = np.asarray(R.TABLE[f"{name}_{fn}"](pts, *params), dtype=np.float64)
    path = os.path.join(ROOT, "tests", "golden", "elemwise.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
