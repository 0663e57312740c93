"""Round-3 debugging of the table kernels on the GPU box (prints; not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
np.set_printoptions(precision=17, linewidth=200)
from oracle import restatement as R
from distributions_amd import _lib
import importlib
import tests.test_gpu_tables_full as T

L = _lib.lib()
def both(name, fn, pts, params):
    m = importlib.import_module(f"distributions_amd.{name}")
    exact = T._exact(L, getattr(m, fn), pts, *params)
    old = L.m355_set_tab_min_n(1)
    try:
        tab = getattr(m, fn)(pts, *params)
    finally:
        L.m355_set_tab_min_n(old)
    with np.errstate(all="ignore"):
        exp = R.TABLE[f"{name}_{fn}"](pts, *params)
    return tab, exact, exp

# 1. gamma(2.5, 1.3): miss fraction by decade of the tail probability
for name, params in (("gamma", (2.5, 1.3)), ("beta", (2.0, 5.0)), ("studentt", (5.0, 0.0, 2.0))):
    for fn, tailfn in (("logcdf", "ppf"), ("logsf", "isf"), ("cdf", "ppf"), ("sf", "isf")):
        row = []
        for lo in range(-9, -1):
            u = np.exp(np.random.default_rng(lo + 20).uniform(np.log(10.0**lo * 1.2), np.log(10.0 ** (lo + 1)), 4096))
            x = R.TABLE[f"{name}_{tailfn}"](u, *params)
            tab, exact, exp = both(name, fn, x, params)
            row.append((lo, round(float(np.mean(tab != exact)), 3), float(np.max(np.abs(tab - exp) / np.abs(exp)))))
        print(name, fn, "decade: frac(tab != exact), max rel err vs oracle:", row, flush=True)

# 2. the sweep's failures, in full
sets = {n: T.box_sets(n) for n in ("gamma", "beta", "studentt")}
per = (T.N_SETS + T.CHUNKS - 1) // T.CHUNKS
for name, chunk in (("beta", 0), ("beta", 1), ("beta", 3)):
    rng = np.random.default_rng(1000 + chunk)
    for params in sets[name][chunk * per:(chunk + 1) * per]:
        x, q = T.box_points(name, params, rng)
        for fn in ("logcdf", "logsf", "cdf", "sf", "ppf", "isf"):
            quant = fn in ("ppf", "isf")
            pts = q if quant else x
            tab, exact, exp = both(name, fn, pts, params)
            fin = np.isfinite(exact)
            with np.errstate(all="ignore"):
                if quant:
                    lg = np.where(exact > 0, np.abs(np.log(np.where(exact > 0, exact, 1.0))), 0.0)
                    bad = fin & ~(np.abs(tab - exact) <= (2e-13 + 3e-14 * lg) * np.maximum(np.abs(exact), 1e-300))
                    import tests.harness as H
                    bad2 = ~(H.ppf_close(name, fn, params, tab, exp) | H.ppf_close(name, fn, params, exp, tab) | H.ppf_unpinned(name, fn, pts) | ~np.isfinite(exp) | (tab == exact))
                else:
                    import tests.harness as H
                    bad = fin & ~(np.abs(tab - exact) <= 5e-13 * np.maximum(1.0, np.abs(exact)))
                    bad2 = ~(H.close(tab, exp) | (tab == exact))
            for label, b in (("vs exact", bad), ("vs oracle", bad2)):
                if b.any():
                    i = np.where(b)[0][:3]
                    print(name, params, fn, label, int(b.sum()), "pts", pts[i], "tab", tab[i], "exact", exact[i], "oracle", exp[i], flush=True)

# 3. beta(0.5, 499.5) logsf around x = 8e-7: table vs exact vs oracle on a dense grid
print("---- beta(0.5, 499.5) logsf dense grid")
xs = np.exp(np.linspace(np.log(2e-7), np.log(4e-6), 4000))
tab, exact, exp = both("beta", "logsf", xs, (0.5, 499.5))
d = np.abs(tab - exact)
import mpmath as mp
mp.mp.dps = 40
for i in np.argsort(-d)[:5]:
    ref = mp.log(mp.betainc(499.5, 0.5, 0, 1 - mp.mpf(xs[i]), regularized=True))
    print("x", xs[i], "tab", tab[i], "exact", exact[i], "oracle", exp[i], "mp(true, smooth)", float(ref), "tab-exact", tab[i] - exact[i])
print("fraction tab != exact", np.mean(tab != exact), "max diff", d.max(), "count > 5e-13", int((d > 5e-13).sum()))
blocks = np.array_split(np.arange(xs.size), 40)
print("per block max |tab-exact| and miss fraction:", [(f"{xs[b[0]]:.2e}", f"{d[b].max():.1e}", round(float(np.mean(tab[b] == exact[b])), 2)) for b in blocks])
