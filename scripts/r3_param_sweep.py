"""Round 3: random-parameter sweep of the six headline distributions' exact kernels against the oracle (prints the worst
cases per (dist, fn); exploratory -- findings become tests)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import numpy as np
from oracle import restatement as R
from tests import harness as H

rng = np.random.default_rng(2026)
def lu(lo, hi, n): return np.exp(rng.uniform(np.log(lo), np.log(hi), n))
NS, NP = 300, 400   # parameter sets x points
def run(name, params, xs, fns):
    m = importlib.import_module(f"distributions_amd.{name}")
    pa = [np.repeat(p, NP) for p in params]
    for fn in fns:
        pts = xs[fn if fn in xs else "x"]
        with np.errstate(all="ignore"):
            got = getattr(m, fn)(pts, *pa)
            exp = R.TABLE[f"{name}_{fn}"](pts, *pa)
            ok = H.ppf_close(name, fn, tuple(pa), got, exp) | H.ppf_unpinned(name, fn, pts)
            err = np.abs(got - exp) / np.maximum(1.0, np.abs(exp))
        bad = ~ok
        msg = f"{name}.{fn}: {int(bad.sum())} of {ok.size} outside the bar"
        if bad.any():
            i = np.argsort(-np.where(bad, np.where(np.isfinite(err), err, 1e300), -1))[:4]
            msg += "; worst: " + "; ".join(f"x={pts[j]!r} params={[float(p[j]) for p in pa]} got={got[j]!r} exp={exp[j]!r}" for j in i)
        print(msg, flush=True)

# poisson
mu = lu(1e-2, 1e6, NS)
k = rng.poisson(np.repeat(np.minimum(mu, 1e9), NP)).astype(np.float64)
q = rng.uniform(0, 1, NS * NP)
run("poisson", (mu,), {"x": k, "ppf": q}, ("logpdf", "cdf", "sf", "logcdf", "logsf", "ppf"))
# binomial
n = np.floor(lu(1, 1e5, NS)); p = rng.uniform(1e-4, 1 - 1e-4, NS)
k = rng.binomial(np.repeat(n, NP).astype(np.int64), np.repeat(p, NP)).astype(np.float64)
run("binomial", (n, p), {"x": k, "ppf": q}, ("logpdf", "cdf", "sf", "logcdf", "logsf", "ppf"))
# gamma
al = lu(1e-2, 1e5, NS); be = lu(1e-3, 1e3, NS)
x = rng.gamma(np.repeat(al, NP), 1 / np.repeat(be, NP))
run("gamma", (al, be), {"x": x, "ppf": q}, ("logpdf", "logcdf", "logsf", "ppf"))
# beta
a = lu(1e-2, 1e4, NS); b = lu(1e-2, 1e4, NS)
x = rng.beta(np.repeat(a, NP), np.repeat(b, NP))
run("beta", (a, b), {"x": x, "ppf": q}, ("logpdf", "logcdf", "logsf", "ppf"))
# studentt
nu = lu(0.1, 1e6, NS); mu_ = rng.normal(0, 10, NS); sg = lu(1e-3, 1e3, NS)
x = np.repeat(mu_, NP) + np.repeat(sg, NP) * rng.standard_t(np.repeat(np.minimum(nu, 1e5), NP))
run("studentt", (nu, mu_, sg), {"x": x, "ppf": q}, ("logpdf", "logcdf", "logsf", "ppf"))
# normal
x = np.repeat(mu_, NP) + np.repeat(sg, NP) * rng.normal(0, 3, NS * NP)
run("normal", (mu_, sg), {"x": x, "ppf": q}, ("logpdf", "logcdf", "logsf", "ppf"))
