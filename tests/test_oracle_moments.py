"""Pins oracle/restatement_moments.py the way the reference pins its moment functions: against scipy.stats on the
parameter grids of the reference's tests, with the tolerances and skips of tests/helper_scipy.py:45-48,163-238 and
of the per-distribution tests (tests/test_<dist>.py: entropy_rtol, skewness_rtol, skip_mode ...).  Families the
reference tests only empirically (no SciPy counterpart: kumaraswamy, logitnormal, discreteweibull, zero-inflated,
rice against scipy.stats.rice where it exists, polyagamma) are pinned on numerical integrals of the oracle's own
pinned pdf."""

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import restatement as R
from oracle import restatement_moments as M
from distributions_amd._lib import DISCRETE  # names only (pure Python)
from tests import harness as H

# keyword arguments the reference's tests pass to run_distribution_tests (tests/test_<dist>.py of the reference)
RTOL = {
    "exgaussian": {"entropy": 1e-1},                                   # test_exgaussian.py:31
    "skewnormal": {"entropy": 1e-2},                                   # test_skewnormal.py:30
    "skew_studentt": {k: 1e-5 for k in ("mean", "var", "std", "skewness", "kurtosis")},  # test_skew_studentt.py:41-45
    "truncatednormal": {"skewness": 1e-4, "kurtosis": 1e-4},           # test_truncatednormal.py:40-41
    "rice": {"mean": 1e-2, "var": 1e-2, "std": 1e-2, "skewness": 1e-1, "kurtosis": 1e-1, "entropy": 1e-2},  # test_rice.py:38-43
}
NB_ENTROPY_RTOL = {(2.1, 0.375): 1e-3, (1.0, 0.8): 1e-2, (10.0, 0.1): 1e-3, (0.5, 0.9): 1e-1, (20.0, 0.05): 1e-1}  # test_negativebinomial.py:13-17
SKIP = {
    "cauchy": {"skewness", "kurtosis"},          # test_cauchy.py:33-34
    "halfstudentt": {"skewness", "kurtosis"},    # test_halfstudentt.py:18-19
    "discreteuniform": {"mode"},                 # test_discreteuniform.py:34
    "skew_studentt": {"mode"},                   # test_skew_studentt.py:35
    "vonmises": {"mode", "std", "var", "skewness", "kurtosis"},  # test_vonmises.py:32-38 (SciPy's are not the circular ones)
}
DEFAULT = {"mean": 1e-6, "median": 1e-6, "var": 1e-6, "std": 1e-6, "skewness": 1e-6, "kurtosis": 1e-6, "entropy": 1e-4}

_GRIDS = H.grids()


def _ids(g):
    return [f"{e[0]}-{'-'.join(str(v) for v in e[2])}" for e in g]


@pytest.mark.parametrize("name, sp_dist, params, sp_params, support, kw", _GRIDS, ids=_ids(_GRIDS))
def test_oracle_moments_vs_scipy(name, sp_dist, params, sp_params, support, kw):
    d = sp_dist(**sp_params)
    rt = dict(DEFAULT)
    rt.update(RTOL.get(name, {}))
    if name == "halfstudentt" and params[0] == 1e4:
        rt.update({"entropy": 1e-2, "median": 1e-2})  # test_halfstudentt.py:30,38 (the nu = 1e4 case)
    if name == "negativebinomial":
        rt["entropy"] = NB_ENTROPY_RTOL[tuple(params)]
    skip = SKIP.get(name, set())
    if name in ("beta", "betabinomial") and not (params[-2] > 1 and params[-1] > 1):
        skip = skip | {"mode"}  # test_beta.py / test_betabinomial.py: skip_mode where no unique interior mode
    with np.errstate(all="ignore"):
        sk, ku = (float(v) for v in d.stats(moments="sk"))
        expected = {"mean": d.mean(), "median": d.median(), "var": d.var(), "std": d.std(), "skewness": sk, "kurtosis": ku,
                    "entropy": d.entropy()}
    for stat, exp in expected.items():
        if stat in skip:
            continue
        got = M.moment(name, stat, *params)
        assert got.shape == (), (name, stat, got.shape)
        assert_allclose(got, exp, rtol=rt[stat], atol=1e-15 if stat == "median" else 0, err_msg=f"{name}.{stat} {params}")
    if "mode" not in skip:  # helper_scipy.py:182-199
        mode = float(M.moment(name, "mode", *params))
        pdf = R.TABLE[f"{name}_pdf"]
        eps = 1.0 if name in DISCRETE else float(np.diff(R.TABLE[f"{name}_ppf"](np.array([0.6, 0.4]), *params))[0]) * 0.01
        with np.errstate(all="ignore"):
            pm, pl, pr = (float(pdf(np.float64(v), *params)) for v in (mode, mode - eps, mode + eps))
        assert pm >= pl - 1e-4 and pm >= pr - 1e-4, (name, params, mode, pm, pl, pr)


def _integrate(name, params, lo, hi, n=400_001):
    x = np.linspace(lo, hi, n)
    with np.errstate(all="ignore"):
        p = R.TABLE[f"{name}_pdf"](x, *params)
    p = np.where(np.isfinite(p), p, 0.0)
    w = np.trapezoid if hasattr(np, "trapezoid") else np.trapz
    m = w(x * p, x)
    c = [w((x - m) ** k * p, x) for k in (2, 3, 4)]
    with np.errstate(all="ignore"):
        ent = -w(np.where(p > 0, p * np.log(np.where(p > 0, p, 1.0)), 0.0), x)
    return {"mean": m, "var": c[0], "std": np.sqrt(c[0]), "skewness": c[1] / c[0] ** 1.5, "kurtosis": c[2] / c[0] ** 2 - 3, "entropy": ent}


@pytest.mark.parametrize("name, params, lo, hi, tol", [
    ("kumaraswamy", (2.0, 5.0), 0.0, 1.0, 1e-6), ("kumaraswamy", (1.5, 3.0), 0.0, 1.0, 1e-6), ("kumaraswamy", (5.0, 1.0), 0.0, 1.0, 2e-4),
    ("logitnormal", (0.0, 1.0), 1e-12, 1 - 1e-12, 1e-6), ("logitnormal", (1.0, 0.5), 1e-12, 1 - 1e-12, 1e-6),
    ("rice", (2.0, 1.0), 1e-10, 14.0, 1e-3), ("rice", (0.5, 2.0), 1e-10, 24.0, 1e-3),
    ("polyagamma", (1.0, 0.5), 1e-10, 6.0, 1e-3), ("polyagamma", (3.0, 2.0), 1e-10, 8.0, 1e-3),
])
def test_oracle_moments_vs_numerical_integral(name, params, lo, hi, tol):
    """No SciPy counterpart (or, for rice, quadrature-defined skewness / kurtosis / entropy): the closed forms and
    1000-point trapezoids of the reference against a 400001-point integral of the pinned pdf."""
    exp = _integrate(name, params, lo, hi)
    for stat in ("mean", "var", "std", "skewness", "kurtosis", "entropy"):
        got = float(M.moment(name, stat, *params))
        # polyagamma: the reference integrates to mean + 10 std only, which cuts ~1 % off the third and fourth moments
        t = 5e-2 if (name == "polyagamma" and stat in ("skewness", "kurtosis")) else tol
        assert abs(got - exp[stat]) <= t * max(1.0, abs(exp[stat])), (name, stat, params, got, exp[stat])


@pytest.mark.parametrize("name, params", [("discreteweibull", (0.5, 0.8)), ("discreteweibull", (0.9, 1.3)), ("zi_poisson", (0.7, 3.0)),
                                          ("zi_poisson", (0.4, 15.0)), ("zi_binomial", (0.6, 12.0, 0.4)),
                                          ("zi_negativebinomial", (0.8, 5.0, 0.4))])
def test_oracle_moments_discrete_vs_support_sum(name, params):
    """Discrete families without a SciPy counterpart: the reference's closed forms / truncated sums against the sum
    over a support long enough to hold all but ~1e-12 of the mass (the reference's own sums truncate at the 1e-4
    quantiles: 1e-2, the tolerance of tests/test_discreteweibull.py:30-32)."""
    k = np.arange(0, 200_000, dtype=np.float64)
    with np.errstate(all="ignore"):
        p = R.TABLE[f"{name}_pdf"](k, *params)
    m = np.sum(k * p)
    c = [np.sum((k - m) ** j * p) for j in (2, 3, 4)]
    exp = {"mean": m, "var": c[0], "std": np.sqrt(c[0]), "skewness": c[1] / c[0] ** 1.5, "kurtosis": c[2] / c[0] ** 2 - 3,
           "entropy": -np.sum(np.where(p > 0, p * np.log(np.where(p > 0, p, 1.0)), 0.0))}
    loose = name == "discreteweibull"
    for stat, e in exp.items():
        got = float(M.moment(name, stat, *params))
        tol = (0.25 if stat in ("skewness", "kurtosis") else 2e-2) if loose else (2e-2 if stat == "entropy" else 1e-9)
        assert abs(got - e) <= tol * max(1.0, abs(e)), (name, stat, params, got, e)
    if name != "discreteweibull":  # discreteweibull.py:19 is a formula in log(beta): negative for beta < 1, kept as written
        mode = float(M.moment(name, "mode", *params))
        assert p[int(mode)] >= p.max() - 1e-12, (name, params, mode)


def test_oracle_moments_broadcast():
    """Arrays of parameters broadcast like every other function of the module (per parameter set)."""
    a = np.array([0.5, 2.0, 30.0])
    b = np.array([[1.0], [4.0]])
    for stat in M.STATS:
        got = M.moment("gamma", stat, a, b)
        assert got.shape == (2, 3)
        for i in range(2):
            for j in range(3):
                np.testing.assert_array_equal(got[i, j], M.moment("gamma", stat, a[j], b[i, 0]))
    assert M.moment("rice", "kurtosis", np.array([1.0, 2.0]), 1.0).shape == (2,)
