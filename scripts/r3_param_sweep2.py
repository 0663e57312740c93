"""Round 3: random-parameter sweep of the gamma / beta relatives (exact kernels) against the oracle (exploratory)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import numpy as np
from oracle import restatement as R
from tests import harness as H

rng = np.random.default_rng(77)
def lu(lo, hi, n): return np.exp(rng.uniform(np.log(lo), np.log(hi), n))
NS, NP = 250, 400
def run(name, params, fns=("logpdf", "logcdf", "logsf", "cdf", "sf", "ppf", "isf")):
    m = importlib.import_module(f"distributions_amd.{name}")
    pa = [np.repeat(p, NP) for p in params]
    with np.errstate(all="ignore"):
        x = np.asarray(m.rvs(*pa, random_state=5), dtype=np.float64)
    q = rng.uniform(0, 1, NS * NP)
    q[::7] = np.exp(rng.uniform(-30, 0, q[::7].size)); q[3::7] = 1 - np.exp(rng.uniform(-30, 0, q[3::7].size))
    for fn in fns:
        pts = q if fn in ("ppf", "isf") else x
        with np.errstate(all="ignore"):
            got = getattr(m, fn)(pts, *pa)
            exp = R.TABLE[f"{name}_{fn}"](pts, *pa)
            ok = H.ppf_close(name, fn, tuple(pa), got, exp) | H.ppf_unpinned(name, fn, pts) | H.family_ok(name, fn, got, exp)
            err = np.abs(got - exp) / np.maximum(1.0, np.abs(exp))
        bad = ~ok
        msg = f"{name}.{fn}: {int(bad.sum())} of {ok.size} outside the bar"
        if bad.any():
            i = np.argsort(-np.where(bad, np.where(np.isfinite(err), err, 1e300), -1))[:3]
            msg += "; worst: " + "; ".join(f"x={pts[j]!r} params={[float(p[j]) for p in pa]} got={got[j]!r} exp={exp[j]!r}" for j in i)
        print(msg, flush=True)

run("inversegamma", (lu(0.05, 3e3, NS), lu(1e-3, 1e3, NS)))
run("chisquared", (lu(0.05, 1e4, NS),))
run("scaledinversechisquared", (lu(0.1, 3e3, NS), lu(1e-3, 1e3, NS)))
lo = rng.normal(0, 10, NS)
run("betascaled", (lu(0.05, 1e3, NS), lu(0.05, 1e3, NS), lo, lo + lu(1e-2, 1e3, NS)))
run("halfstudentt", (lu(0.2, 1e5, NS), lu(1e-3, 1e3, NS)))
run("weibull", (lu(0.05, 50, NS), lu(1e-3, 1e3, NS)))
run("lognormal", (rng.normal(0, 3, NS), lu(1e-3, 5, NS)))
run("loglogistic", (lu(1e-3, 1e3, NS), lu(0.1, 50, NS)))
run("negativebinomial", (lu(0.1, 1e3, NS), rng.uniform(0.01, 0.99, NS)), fns=("logpdf", "logcdf", "logsf", "cdf", "sf", "ppf"))
