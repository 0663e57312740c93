"""The table kernels (k_tab, k_coltab: csrc/m355_elemwise.h) at the size and in the parameter box they serve.

1. BASELINE.json's config 2 at its full size, N = 1e8 fp64, through the DEFAULT dispatch (no forced threshold): gamma /
   beta / studentt logcdf, logsf, ppf with scalar parameters and with (1024,) row-periodic parameters (config 2's
   linspace, bench_configs.py); every 97th element against the oracle at 1e-12, the last elements of the array, and
   `mean(table != exact kernel) > 0.5` on the sample to show that the table kernels really ran.
   Reference formulas: gamma.py:65-94, 48-50; beta.py:90-120, 78-79; studentt.py:79-105.
2. A randomized sweep over each `Dist<>::tab_ok` acceptance box INCLUDING ITS EDGES (gamma alpha in {0.01, 1e4}, beta
   a + b = 500, studentt nu in {1, 50}, rate / scale at 1e-150 and 1e150), 20 000 points per parameter set, forced
   threshold: against the exact kernels and the oracle.
3. Relative accuracy of tabulated cdf / sf on tail points (p from 1e-9 to 1e-3): the reference composes them as
   exp(logcdf) / exp(logsf) (gamma.py:40-41, beta.py:66-67, 82-83), so a small probability carries the relative accuracy
   of its logarithm; the tables are routed through the log function (Dist<>::tab_route) to keep that.
"""

import numpy as np
import pytest

from oracle import restatement as R
from tests import harness as H

pytestmark = pytest.mark.gpu


def mod(name):
    import importlib

    return importlib.import_module(f"distributions_amd.{name}")


SCALAR = {"gamma": (2.5, 1.3), "beta": (2.0, 5.0), "studentt": (5.0, 0.0, 2.0)}
SAMPLER = {"gamma": ("gamma", (2.5, 1.3), 2), "beta": ("beta", (2.0, 5.0), 3), "studentt": ("studentt", (5.0, 0.0, 2.0), 4)}
P = 1024


def periodic_params(name):
    lin = lambda lo, hi: np.linspace(lo, hi, P)
    return {"gamma": (lin(0.5, 9.0), lin(0.3, 3.0)), "beta": (lin(0.5, 9.0), lin(1.0, 10.0)),
            "studentt": (lin(2.0, 30.0), 0.0, 2.0)}[name]


def _exact(L, f, *args):
    old = L.m355_set_tab_min_n(2**62)
    try:
        return f(*args)
    finally:
        L.m355_set_tab_min_n(old)


@pytest.mark.parametrize("name", ["gamma", "beta", "studentt"])
def test_full_size_1e8_table_kernels(gpu, name):
    import torch

    from distributions_amd import _lib, raw

    L = _lib.lib()
    assert L.m355_set_tab_min_n(-1) == 2**21  # the default dispatch (and leave it there)
    m = mod(name)
    rows = 100_000_000 // P
    n = rows * P  # 99 999 744: the (rows, 1024) view of config 2's broadcast layout
    sampler, sp, seed = SAMPLER[name]
    x = torch.empty(n, dtype=torch.float64, device=gpu)
    raw.rvs_into(sampler, sp, x, seed=seed)
    q = torch.empty(n, dtype=torch.float64, device=gpu)
    raw.rvs_into("uniform", (0.0, 1.0), q, seed=5)
    q.clamp_(1e-12, 1 - 1e-12)
    idx = torch.arange(0, n, 97, device=gpu)
    tail = torch.arange(n - 5000, n, device=gpu)
    pick = torch.cat([idx, tail])
    col = (pick % P).cpu().numpy()
    pp = periodic_params(name)
    pp_t = tuple(torch.from_numpy(v).to(gpu) if isinstance(v, np.ndarray) else v for v in pp)
    for fn in ("logcdf", "logsf", "ppf"):
        src = q if fn == "ppf" else x
        pts = src[pick].cpu().numpy()
        for layout in ("scalar", "periodic"):
            if layout == "scalar":
                params = SCALAR[name]
                out = getattr(m, fn)(src, *params)
                elem_params = params
            else:
                out = getattr(m, fn)(src.view(rows, P), *pp_t).view(-1)
                elem_params = tuple(v[col] if isinstance(v, np.ndarray) else v for v in pp)
            got = out[pick].cpu().numpy()
            del out
            with np.errstate(all="ignore"):
                exp = R.TABLE[f"{name}_{fn}"](pts, *elem_params)
            ok = H.ppf_close(name, fn, elem_params, got, exp) | H.ppf_unpinned(name, fn, pts)
            assert np.all(ok), (name, fn, layout, int((~ok).sum()), pts[~ok][:4], got[~ok][:4], exp[~ok][:4])
            # the exact kernels on the same points (full-size parameter arrays for the periodic layout: k_generic)
            exact = _exact(L, getattr(m, fn), pts, *elem_params)
            assert np.all(H.ppf_close(name, fn, elem_params, got, exact) | H.ppf_close(name, fn, elem_params, exact, got)), (
                name, fn, layout)
            assert np.mean(got != exact) > 0.5, (name, fn, layout, float(np.mean(got != exact)))  # the table kernel ran


# ---------------------------------------------------------------------------------------------------------------------
# 2. the acceptance boxes, edges included

N_SETS = 208
CHUNKS = 4
N_PTS = 20_000


def _loguniform(rng, lo, hi, size=None):
    return np.exp(rng.uniform(np.log(lo), np.log(hi), size))


def box_sets(name):
    """>= 200 parameter sets: the box's corners and edges first, then log-uniform draws over it."""
    rng = np.random.default_rng({"gamma": 101, "beta": 102, "studentt": 103}[name])
    sets = []
    if name == "gamma":  # Dist<M355_GAMMA>::tab_ok: alpha in [0.01, 1e4], beta in (0, inf)
        for a in (0.01, 0.05, 0.999, 1.0, 1e4):
            for b in (1e-150, 1.0, 1e150):
                sets.append((a, b))
        sets += [(1e4, 3.7), (9999.5, 1e-3), (0.0100001, 2.0), (0.05, 1e-150), (0.049999, 1.0)]
        while len(sets) < N_SETS:
            sets.append((float(_loguniform(rng, 0.01, 1e4)), float(_loguniform(rng, 1e-150, 1e150))))
    elif name == "beta":  # a, b >= 0.01, a + b <= 500
        sets += [(0.01, 0.01), (0.01, 499.99), (499.99, 0.01), (250.0, 250.0), (1.0, 499.0), (499.0, 1.0), (0.5, 499.5),
                 (499.5, 0.5), (100.0, 400.0), (400.0, 100.0), (0.05, 1.0), (0.05, 499.95), (0.5, 0.5), (0.999, 1.0),
                 (1.0, 0.999), (0.01, 1.0), (1.0, 0.01), (30.0, 470.0), (0.049, 3.0), (3.0, 0.99)]
        while len(sets) < N_SETS:
            a, b = float(_loguniform(rng, 0.01, 500.0)), float(_loguniform(rng, 0.01, 500.0))
            if a + b > 500.0:  # onto the edge a + b = 500
                s = 500.0 / (a + b)
                a, b = max(a * s, 0.01), max(b * s, 0.01)
                if a + b > 500.0:
                    a, b = (a - (a + b - 500.0), b) if a > b else (a, b - (a + b - 500.0))
            sets.append((a, b))
    else:  # studentt: nu in [1, 50], sigma in (0, inf), mu finite
        for nu in (1.0, 1.0000001, 2.0, 49.999, 50.0):
            for sg in (1e-150, 1.0, 1e150):
                sets.append((nu, 0.0, sg))
        sets += [(1.0, 1e6, 1e-3), (50.0, -1e10, 2.0), (50.0, 3.0, 1e150), (1.0, -2.0, 1e-150), (7.0, 1e150, 1e150)]
        while len(sets) < N_SETS:
            sg = float(_loguniform(rng, 1e-150, 1e150))
            mu = float(rng.normal() * sg * rng.choice([0.0, 1.0, 1e3]))
            sets.append((float(_loguniform(rng, 1.0, 50.0)), mu, sg))
    return sets[:N_SETS]


def box_points(name, params, rng):
    """Draws of the distribution itself plus both tails inside the table's range (quantiles 1e-9 ... 1e-3)."""
    n_tail = 1500
    with np.errstate(all="ignore"):
        u = np.exp(rng.uniform(np.log(1.2e-9), np.log(1e-3), n_tail))
        if name == "gamma":
            x = rng.gamma(params[0], 1.0, N_PTS) / params[1]
        elif name == "beta":
            x = rng.beta(params[0], params[1], N_PTS)
        else:
            x = params[1] + params[2] * rng.standard_t(params[0], N_PTS)
        lo_t = R.TABLE[f"{name}_ppf"](u, *params)
        hi_t = R.TABLE[f"{name}_isf"](u, *params)
    x[:n_tail] = lo_t
    x[n_tail : 2 * n_tail] = hi_t
    q = rng.uniform(0, 1, N_PTS)
    q[:n_tail] = u
    q[n_tail : 2 * n_tail] = 1.0 - u
    return x, q


def tab_ok_fn(name, fn, params):
    """Dist<>::tab_ok_fn of csrc/m355_dists.h: which functions of an accepted parameter set have a table."""
    if name == "gamma":
        return fn not in ("ppf", "isf") or params[0] >= 0.05
    if name == "beta":
        if fn in ("ppf", "isf"):
            return params[0] >= 0.1 and params[1] >= 1.0
        if fn in ("logsf", "sf"):
            return params[0] >= 0.5
    return True


@pytest.mark.parametrize("chunk", range(CHUNKS))
@pytest.mark.parametrize("name", ["gamma", "beta", "studentt"])
def test_tab_acceptance_box_sweep(gpu, name, chunk):
    from distributions_amd import _lib

    L = _lib.lib()
    m = mod(name)
    sets = box_sets(name)
    per = (len(sets) + CHUNKS - 1) // CHUNKS
    rng = np.random.default_rng(1000 + chunk)
    worst = {}
    for params in sets[chunk * per : (chunk + 1) * per]:
        x, q = box_points(name, params, rng)
        for fn in ("logcdf", "logsf", "cdf", "sf", "ppf", "isf"):
            quant = fn in ("ppf", "isf")
            pts = q if quant else x
            exact = _exact(L, getattr(m, fn), pts, *params)
            old = L.m355_set_tab_min_n(1)
            try:
                tab = getattr(m, fn)(pts, *params)
            finally:
                L.m355_set_tab_min_n(old)
            assert np.array_equal(np.isnan(tab), np.isnan(exact)), (name, params, fn)
            fin = np.isfinite(exact)
            assert np.array_equal(tab[~fin], exact[~fin], equal_nan=True), (name, params, fn)
            with np.errstate(all="ignore"):
                exp = R.TABLE[f"{name}_{fn}"](pts, *params)
            if quant:
                with np.errstate(all="ignore"):  # (tables of log(ppf): relative error ~1e-14 max(1, |log x|))
                    lg = np.where(exact > 0, np.abs(np.log(np.where(exact > 0, exact, 1.0))), 0.0)
                scale = np.abs(exact - (params[1] if name == "studentt" else 0.0))  # (relative to the root's distance from mu)
                err = np.abs(tab - exact)
                tol = (2e-13 + 3e-14 * lg) * np.maximum(scale, 1e-300)
                bad = fin & ~(err <= tol)
                if name == "studentt":  # the exact body follows the reference's quantised sqrt(1/b - 1): modelled window
                    bad &= ~(H.ppf_close(name, fn, params, tab, exact) | H.ppf_close(name, fn, params, exact, tab))
                # the oracle: the parity bar, with |mu| in its scale for studentt (x = mu + r sigma rounds at 1.1e-16 max(|mu|,
                # |x|): next to a zero crossing of the quantile max(1, |x|) alone is not the scale of anything)
                ok_t = H.ppf_close(name, fn, params, tab, exp) | H.ppf_close(name, fn, params, exp, tab)
                ok_e = H.ppf_close(name, fn, params, exact, exp) | H.ppf_close(name, fn, params, exp, exact)
                if name == "studentt":
                    ok_t |= np.abs(tab - exp) <= 1e-12 * np.maximum(np.abs(exp), abs(params[1]))
                    ok_e |= np.abs(exact - exp) <= 1e-12 * np.maximum(np.abs(exp), abs(params[1]))
                ok_or = ok_t | H.ppf_unpinned(name, fn, pts) | ~np.isfinite(exp) | (tab == exact)
            else:
                err = np.abs(tab - exact)
                bad = fin & ~(err <= 5e-13 * np.maximum(1.0, np.abs(exact)))
                ok_or = H.close(tab, exp) | (tab == exact)
                ok_e = H.close(exact, exp)
            # the exact kernels' own parity is the business of tests/test_gpu_elemwise.py and test_gpu_regimes.py (and where
            # SciPy itself is off -- betainc next to x = 1 at shapes below 1, 3e-12 at beta(1/2, 1/2) against the closed form
            # (2 / pi) asin(sqrt x) -- both kernels leave it together): the table must not ADD a miss against the oracle
            ok_or |= ~ok_e
            assert not bad.any(), (name, params, fn, "vs exact kernel", int(bad.sum()), pts[bad][:3], tab[bad][:3], exact[bad][:3])
            assert np.all(ok_or), (name, params, fn, "vs oracle", int((~ok_or).sum()), pts[~ok_or][:3], tab[~ok_or][:3], exp[~ok_or][:3])
            if tab_ok_fn(name, fn, params):
                worst.setdefault(fn, []).append(float(np.mean(tab != exact)))
    # over the chunk the tables were in use (single sets may legitimately miss a lot: poisoned cells at the box's edges)
    for fn, fr in worst.items():
        assert np.median(fr) > 0.5, (name, fn, fr)


# ---------------------------------------------------------------------------------------------------------------------
# 3. relative accuracy of tabulated cdf / sf in the tails

REL_CASES = [("gamma", (2.5, 1.3)), ("gamma", (0.5, 2.0)), ("gamma", (50.0, 0.5)), ("gamma", (3000.0, 7.0)),
             ("beta", (2.0, 5.0)), ("beta", (0.5, 3.0)), ("beta", (100.0, 100.0)), ("beta", (30.0, 1.5)),
             ("studentt", (5.0, 0.0, 2.0)), ("studentt", (1.0, 1.0, 0.5)), ("studentt", (30.0, -3.0, 1e-3)),
             ("inversegamma", (3.0, 2.0)), ("chisquared", (7.0,)), ("scaledinversechisquared", (5.0, 0.7)),
             ("betascaled", (2.0, 5.0, -1.0, 3.0))]


@pytest.mark.parametrize("name, params", REL_CASES, ids=[f"{n}-{'-'.join(str(v) for v in p)}" for n, p in REL_CASES])
def test_tabulated_cdf_sf_relative_accuracy(gpu, name, params):
    """Tail points with probabilities from 1e-9 to 1e-3: wherever the exact kernel's cdf / sf agrees with the oracle to
    1e-13 RELATIVE, the tabulated one must agree to 1e-12 relative (it is exp of the tabulated log function)."""
    from distributions_amd import _lib

    L = _lib.lib()
    m = mod(name)
    rng = np.random.default_rng(77)
    n = 30_000
    u = np.exp(rng.uniform(np.log(1.2e-9), np.log(1e-3), n))
    with np.errstate(all="ignore"):
        pts = {"cdf": R.TABLE[f"{name}_ppf"](u, *params), "sf": R.TABLE[f"{name}_isf"](u, *params)}
    for fn in ("cdf", "sf"):
        x = pts[fn]
        exact = _exact(L, getattr(m, fn), x, *params)
        old = L.m355_set_tab_min_n(1)
        try:
            tab = getattr(m, fn)(x, *params)
        finally:
            L.m355_set_tab_min_n(old)
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"{name}_{fn}"](x, *params)
            rel_exact = np.abs(exact - exp) / np.abs(exp)
            rel_tab = np.abs(tab - exp) / np.abs(exp)
        held = rel_exact <= 1e-13
        if fn == "cdf" or name in ("beta", "studentt", "inversegamma", "scaledinversechisquared"):
            assert held.mean() > 0.5, (name, fn, float(held.mean()))  # (the exact kernel itself: 1e-13 relative mostly)
            assert np.all(rel_tab[held] <= 1e-12), (name, fn, float(rel_tab[held].max()), x[held][np.argmax(rel_tab[held])])
        else:
            # sf = 1 - cdf in the reference of gamma / chisquared / betascaled (gamma.py:53-54, chisquared.py:77-78,
            # betascaled.py:88-89): a small sf is a rounded difference -- absolute accuracy only, in the oracle, the exact
            # kernel and the table alike
            assert np.all(np.abs(tab - exp) <= 1e-14), (name, fn, float(np.abs(tab - exp).max()))
        if fn == "cdf":  # the table was used (sf = 1 - exp(logcdf) next to 1 rounds to the same double either way)
            assert np.mean(tab != exact) > 0.5


@pytest.mark.parametrize("name, fn", [("gamma", "logcdf"), ("beta", "ppf"), ("studentt", "logsf")])
def test_in_place_row_periodic(gpu, name, fn):
    """out == x with row-periodic parameters and a forced small table threshold (the k_coltab shape): the in-place result
    equals the out-of-place one bit for bit (include/m355_dist.h: x and out may coincide, partial overlap is undefined)."""
    import torch

    from distributions_amd import _lib, raw

    L = _lib.lib()
    rows = 4096
    pp = tuple(torch.from_numpy(v).to(gpu) if isinstance(v, np.ndarray) else v for v in periodic_params(name))
    sampler, sp, seed = SAMPLER[name]
    x = torch.empty(rows * P, dtype=torch.float64, device=gpu)
    raw.rvs_into("uniform" if fn == "ppf" else sampler, (0.0, 1.0) if fn == "ppf" else sp, x, seed=seed)
    x[::1001] = float("nan")
    x[5::1013] = -1.0
    out = torch.empty_like(x)
    ews = raw.elemwise_workspace(gpu)
    old = L.m355_set_tab_min_n(1)
    try:
        raw.elemwise_into(name, fn, x, pp, out, ews)
        y = x.clone()
        raw.elemwise_into(name, fn, y, pp, y, ews)
    finally:
        L.m355_set_tab_min_n(old)
    torch.cuda.synchronize()
    a, b = out.cpu().numpy(), y.cpu().numpy()
    exact = torch.empty_like(x)
    raw.elemwise_into(name, fn, x, pp, exact)  # no workspace: the exact kernels
    assert np.mean(a != exact.cpu().numpy()) > 0.5  # (the per-column tables were in use)
    assert np.array_equal(a, b, equal_nan=True), (name, fn, int(np.sum(~((a == b) | (np.isnan(a) & np.isnan(b))))))
    col = np.arange(a.size) % P
    ep = tuple(v[col] if isinstance(v, np.ndarray) else v for v in periodic_params(name))
    with np.errstate(all="ignore"):
        exp = R.TABLE[f"{name}_{fn}"](x.cpu().numpy(), *ep)
    assert np.all(H.ppf_close(name, fn, ep, a, exp) | H.ppf_unpinned(name, fn, x.cpu().numpy()))


def _partials_written(L, name, fn, x, params, gpu):
    import ctypes

    import torch

    from distributions_amd import _lib
    from distributions_amd._lib import DIST, F64, FN, Operand

    xd = torch.from_numpy(x).to(gpu)
    ops = (Operand * (1 + len(params)))()
    ops[0].ptr, ops[0].scalar, ops[0].div, ops[0].mod, ops[0].dtype = xd.data_ptr(), 0.0, 1, xd.numel(), F64
    for j, p in enumerate(params, start=1):
        ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = None, float(p), 1, 1, F64
    ws = torch.full((int(L.m355_sumlogp_workspace_bytes()) // 8,), float("nan"), dtype=torch.float64, device=gpu)
    out = torch.zeros(1, dtype=torch.float64, device=gpu)
    pp = ctypes.cast(ctypes.byref(ops, ctypes.sizeof(Operand)), ctypes.POINTER(Operand))
    _lib.check(L.m355_sumlogp(DIST[name], FN[fn], xd.numel(), ops, len(params), pp, out.data_ptr(), ws.data_ptr(),
                              ws.numel() * 8, torch.cuda.current_stream(gpu).cuda_stream))
    torch.cuda.synchronize()
    return int((~torch.isnan(ws[:4096])).sum().item())


@pytest.mark.parametrize("name", ["gamma", "beta", "studentt", "normal", "chisquared"])
def test_fused_sum_logcdf_logsf(gpu, name):
    """sum_logcdf / sum_logsf (censored likelihoods; the reference spells them pt.sum(logcdf(...)): gamma.py:65-94,
    beta.py:90-120, studentt.py:100-123, normal.py:70-84): the fused sum against math.fsum of the oracle's terms at
    1e-12 * sum |term| -- above the table threshold (k_tab with the reduction epilogue for the tabulated families), below
    it (exact bodies) and with row-periodic parameters."""
    import math

    from distributions_amd import _lib

    L = _lib.lib()
    m = mod(name)
    rng = np.random.default_rng(91)
    params = {"gamma": (2.5, 1.3), "beta": (2.0, 5.0), "studentt": (5.0, 0.0, 2.0), "normal": (0.3, 1.7), "chisquared": (7.0,)}[name]
    draw = {"gamma": lambda n: rng.gamma(2.5, 1 / 1.3, n), "beta": lambda n: rng.beta(2, 5, n),
            "studentt": lambda n: 2 * rng.standard_t(5, n), "normal": lambda n: rng.normal(0, 1, n),
            "chisquared": lambda n: rng.chisquare(7.0, n)}[name]
    for n in (3_000_017, 40_003):
        x = draw(n)
        for fn in ("logcdf", "logsf"):
            with np.errstate(all="ignore"):
                terms = R.TABLE[f"{name}_{fn}"](x, *params)
            ref, scale = math.fsum(terms), float(np.abs(terms).sum())
            got = getattr(m, f"sum_{fn}")(x, *params)
            assert abs(got - ref) <= 1e-12 * scale, (name, fn, n, got, ref)
            exact = _exact(L, getattr(m, f"sum_{fn}"), x, *params)
            assert abs(exact - ref) <= 1e-12 * scale, (name, fn, n, "exact bodies", exact, ref)
            if n > 2**21 and name != "normal":
                # the table kernel ran: its grid (1280 workgroups) leaves 1280 partials in the workspace, the exact
                # kernel's one per tile of 1024 elements (capped at 4096)
                assert _partials_written(L, name, fn, x, params, gpu) == 1280, (name, fn)
    if name in ("gamma", "beta", "studentt"):  # row-periodic parameters: the exact bodies with the reduction epilogue
        x = draw(64 * P).reshape(64, P)
        pp = periodic_params(name)
        for fn in ("logcdf", "logsf"):
            with np.errstate(all="ignore"):
                terms = R.TABLE[f"{name}_{fn}"](x, *pp)
            got = getattr(m, f"sum_{fn}")(x, *pp)
            assert abs(got - math.fsum(terms.ravel())) <= 1e-12 * float(np.abs(terms).sum()), (name, fn, "periodic")
