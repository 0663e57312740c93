"""Harness-A semantics of the reference's SciPy oracle tests (tests/helper_scipy.py:8-240 in the
reference), restated for eager functions.  Used twice: on the CPU oracle (pinning, no GPU) and on
the HIP product path (``-m gpu``).

The evaluation points are the reference's ``extended_vals``: 20 SciPy draws with
``default_rng(1)`` (which the reference asserts equal its own rvs, helper_scipy.py:51-73),
the two support ends, support[0]-1, support[0]-2, support[1]+1, support[1]+2.
"""

from __future__ import annotations

import types

import numpy as np
from numpy.testing import assert_allclose

from oracle import restatement as R

ISF_POINTS = np.array([-1, 0, 0.25, 0.5, 0.75, 1, 2], dtype=np.float64)


def oracle_namespace(prefix):
    """Expose oracle functions ``<prefix>_<fn>`` as a module-like object (``ns.logpdf`` ...)."""
    ns = types.SimpleNamespace()
    for name, fn in R.TABLE.items():
        if name.startswith(prefix + "_"):
            setattr(ns, name[len(prefix) + 1 :], fn)
    return ns


def extended_vals(scipy_dist, support):
    draws = scipy_dist.rvs(20, random_state=np.random.default_rng(1))
    return np.concatenate(
        [
            np.asarray(draws, dtype=np.float64),
            np.asarray(support, dtype=np.float64),
            [support[0] - 1],
            [support[0] - 2],
            [support[1] + 1],
            [support[1] + 2],
        ]
    )


def run_distribution_checks(dist, sp_dist, params, sp_params, support, rtol=1e-6, logsf_rtol=None,
                            skip_isf=False, to_numpy=np.asarray):
    """pdf/logpdf/cdf/logcdf/sf/logsf on extended_vals and isf on ISF_POINTS vs scipy.stats,
    tolerances exactly as helper_scipy.py:86-161."""
    scipy_dist = sp_dist(**sp_params)
    xs = extended_vals(scipy_dist, support)
    info = f"params={params} scipy={sp_params}"

    def sp_pdf(v):
        return scipy_dist.pdf(v) if hasattr(scipy_dist, "pdf") else scipy_dist.pmf(v)

    def sp_logpdf(v):
        return scipy_dist.logpdf(v) if hasattr(scipy_dist, "logpdf") else scipy_dist.logpmf(v)

    with np.errstate(all="ignore"):
        if hasattr(dist, "pdf"):
            assert_allclose(to_numpy(dist.pdf(xs, *params)), sp_pdf(xs), rtol=rtol,
                            err_msg=f"pdf {info}")
        assert_allclose(to_numpy(dist.logpdf(xs, *params)), sp_logpdf(xs), rtol=rtol,
                        err_msg=f"logpdf {info}")
        if hasattr(dist, "cdf"):
            assert_allclose(to_numpy(dist.cdf(xs, *params)), scipy_dist.cdf(xs), rtol=rtol,
                            atol=1e-12, err_msg=f"cdf {info}")
        if hasattr(dist, "logcdf"):
            assert_allclose(to_numpy(dist.logcdf(xs, *params)), scipy_dist.logcdf(xs), rtol=rtol,
                            atol=1e-12, err_msg=f"logcdf {info}")
        if hasattr(dist, "sf"):
            assert_allclose(to_numpy(dist.sf(xs, *params)), scipy_dist.sf(xs), rtol=rtol,
                            err_msg=f"sf {info}")
        if hasattr(dist, "logsf"):
            assert_allclose(to_numpy(dist.logsf(xs, *params)), scipy_dist.logsf(xs),
                            rtol=logsf_rtol or rtol, err_msg=f"logsf {info}")
        if hasattr(dist, "isf") and not skip_isf:
            assert_allclose(to_numpy(dist.isf(ISF_POINTS, *params)), scipy_dist.isf(ISF_POINTS),
                            rtol=rtol, atol=1e-15, err_msg=f"isf {info}")


# The reference's parameter grids (tests/test_<dist>.py in the reference, lines cited in SURVEY §4).
def grids():
    from scipy import stats

    inf = float("inf")
    g = []
    for mu, sigma in [(0.0, 1.0), (-5.0, 0.5), (-1e6, 100.0), (10.0, 1e-6), (1.0, 1e-6)]:
        g.append(("normal", stats.norm, [mu, sigma], {"loc": mu, "scale": sigma}, (-inf, inf), {}))
    for a, b in [(2.0, 1 / 3), (1.0, 1.0), (0.5, 2.0), (50.0, 0.5), (0.01, 0.01), (100.0, 100.0)]:
        g.append(("gamma", stats.gamma, [a, b], {"a": a, "scale": 1.0 / b}, (0, inf), {}))
    for a, b in [(2.0, 5.0), (0.5, 3.0), (100.0, 100.0), (1.0, 1.0)]:
        g.append(("beta", stats.beta, [a, b], {"a": a, "b": b}, (0, 1), {}))
    for nu, mu, sigma in [(5.0, 0.0, 2.0), (100.0, 0.0, 1.0), (5.0, -2.0, 3.0)]:
        g.append(("studentt", stats.t, [nu, mu, sigma], {"df": nu, "loc": mu, "scale": sigma},
                  (-inf, inf), {}))
    for mu in [2.0, 0.01, 100.0]:
        g.append(("poisson", stats.poisson, [mu], {"mu": mu}, (0, inf), {}))
    for n, p, lrt in [(4, 0.4, 1e-6), (10, 0.3, 1e-6), (100, 0.01, 1e-6), (20, 0.8, 1e-2)]:
        g.append(("binomial", stats.binom, [n, p], {"n": n, "p": p}, (0, n),
                  {"logsf_rtol": lrt}))
    return g


def grid_ids():
    return [f"{name}-{'-'.join(str(v) for v in params)}" for name, _, params, _, _, _ in grids()]


def close(got, exp, tol=1e-12):
    """Parity criterion of BASELINE.md / SURVEY.md 8(d): |a-b| <= tol*max(1,|b|); NaN matches NaN,
    +-inf must match exactly.  Returns the boolean mask."""
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    same = (got == exp) | (np.isnan(got) & np.isnan(exp))
    with np.errstate(all="ignore"):
        return same | (np.abs(got - exp) <= tol * np.maximum(1.0, np.abs(exp)))


def ppf_unpinned(name, fn, pts):
    """Points where the oracle's own inverse (SciPy betaincinv) is known to clamp or return NaN
    ('version-dependent, unpinned'): extreme tails of beta/studentt ppf."""
    if fn in ("ppf", "isf") and name in ("beta", "studentt"):
        q = 1.0 - pts if fn == "isf" else pts
        return (q < 1e-100) | (q > 1 - 1e-15)
    return np.zeros(np.shape(pts), dtype=bool)
