"""Generates tests/golden/moments.json from the CPU oracle (oracle/restatement_moments.py, pinned against scipy.stats
with the reference's tolerances by tests/test_oracle_moments.py): the eight moment functions at every parameter
point of the reference's tests plus the branch points of tests/harness.py::moment_extra_cases.  The reference cannot
be imported here (PyTensor absent).  Run from the repo root:
    python tests/golden/make_golden_moments.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import restatement_moments as M  # noqa: E402
from tests import harness as H  # noqa: E402


def main():
    out = []
    for name, _sp, params, _spp, _sup, _kw in H.grids() + H.moment_extra_cases():
        vals = {st: float(M.moment(name, st, *params)).hex() for st in M.STATS}  # hex: exact, NaN / inf included
        out.append({"dist": name, "params": [float(p).hex() for p in params], "values": vals})
    path = os.path.join(ROOT, "tests", "golden", "moments.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print(path, os.path.getsize(path), "bytes,", len(out), "parameter points")


if __name__ == "__main__":
    main()
