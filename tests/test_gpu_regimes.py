"""The device code in the regimes the g++ host build sees in tests/test_hostmath.py (that build swaps in libm and host
reciprocals, so it is not the device code): wide (a, x) sweeps of gammainc / gammaincc / betainc and their inverses
through gamma / beta .logcdf / .logsf / .ppf with full-size parameter arrays, SURVEY.md Appendix C's edge table as
literal expectations, the studentt nu > 1e5 arm (quirk Q1), and BASELINE.json configs[0] exactly as specified."""

import math

import numpy as np
import pytest
import scipy.special as sp

from oracle import restatement as R
from tests import harness as H

pytestmark = pytest.mark.gpu
inf, nan = np.inf, np.nan


def _log_close(got_log, exp_log, tol):
    return H.close(got_log, exp_log, tol)


@pytest.mark.parametrize("arange", [(1e-3, 1), (1, 10), (10, 100), (100, 1e4)])
def test_device_gammainc_regimes(gpu, arange):
    """gamma.logcdf / logsf with beta = 1: log P(a, x) where x < a and log Q(a, x) where x >= a (gamma.py:65-74,
    85-94; the other side of each is log1p(-Q) / log1p(-P)) -- the sweep of test_host_gammainc on the GPU.  (gamma.sf
    itself is 1 - cdf in the reference, gamma.py:53-54, so it is no test of Q.)  SciPy's own igam loses ~1e-11 for
    a > 100 far from the mean (mpmath check below and in tests/test_hostmath.py), hence the looser SciPy comparison
    there -- the same tolerances as the host test."""
    from distributions_amd import gamma

    rng = np.random.default_rng(1)
    a = np.exp(rng.uniform(np.log(arange[0]), np.log(arange[1]), 20000))
    x = a * np.exp(rng.normal(0, 1.0, a.size))
    tol = 1e-12 if arange[1] <= 100 else 1e-10
    with np.errstate(all="ignore"):
        lo = x < a
        for fn, f, side in (("logcdf", sp.gammainc, lo), ("logsf", sp.gammaincc, ~lo)):
            got = getattr(gamma, fn)(x, a, 1.0)
            direct = f(a, x)
            m = side & (direct > 1e-300)  # subnormal results carry fewer bits
            assert m.sum() > 5000
            ok = H.close(got[m], np.log(direct[m]), tol)
            assert np.all(ok), (fn, arange, a[m][~ok][:4], x[m][~ok][:4], got[m][~ok][:4], np.log(direct[m][~ok][:4]))
            # all points, both arms, against the oracle's two-branch form (same SciPy functions underneath)
            exp = R.TABLE[f"gamma_{fn}"](x, a, 1.0)
            fin = np.isfinite(exp) & (exp > -690)
            ok = H.close(got[fin], exp[fin], tol)
            assert np.all(ok), (fn, arange, a[fin][~ok][:4], x[fin][~ok][:4], got[fin][~ok][:4], exp[fin][~ok][:4])
            assert np.all(np.isneginf(got[np.isneginf(exp)]) | (got[np.isneginf(exp)] < -700))


def test_device_gammainc_vs_mpmath(gpu):
    import mpmath as mp

    from distributions_amd import gamma

    mp.mp.dps = 40
    rng = np.random.default_rng(11)
    a = np.exp(rng.uniform(np.log(0.05), np.log(5e3), 60))
    x = a * np.exp(rng.normal(0, 0.7, a.size))
    lp, lq = gamma.logcdf(x, a, 1.0), gamma.logsf(x, a, 1.0)
    for ai, xi, pi, qi in zip(a, x, lp, lq):
        tp = mp.gammainc(ai, 0, xi, regularized=True)
        tq = mp.gammainc(ai, xi, mp.inf, regularized=True)
        if xi < ai and tp > 1e-290:  # the arm that takes log P directly
            assert abs(pi - mp.log(tp)) <= 1e-12 * max(1, abs(mp.log(tp))), (ai, xi)
        if xi >= ai and tq > 1e-290:  # log Q directly
            assert abs(qi - mp.log(tq)) <= 1e-12 * max(1, abs(mp.log(tq))), (ai, xi)


def test_device_gammainc_temme_region(gpu):
    """a >= 25, |x - a| <= 0.3 a on the device (gammainc_temme; the host twin is test_host_gammainc_temme_region, whose
    docstring says why SciPy stops being the reference at a > 1e5): gamma.logcdf / logsf with beta = 1 against SciPy
    for a <= 1e5 and against the exact-coefficient 60-digit expansion for a up to 1e14, the quantile against SciPy
    (a <= 1e5) and through the forward function (a <= 1e12); poisson.logcdf at mu up to 1e9 (poisson.py:71-75:
    gammaincc(floor(x) + 1, mu)) against the same expansion."""
    import os
    import sys

    import mpmath as mp

    from distributions_amd import gamma, poisson

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import gen_temme_coeffs as gen

    def points(rng, n, lo, hi):
        a = np.exp(rng.uniform(np.log(lo), np.log(hi), n))
        sig = np.where(rng.uniform(size=n) < 0.5, rng.uniform(-0.3, 0.3, n), rng.normal(0, 2.0, n) / np.sqrt(a))
        return a, a * (1 + np.clip(sig, -0.3, 0.3))

    rng = np.random.default_rng(21)
    a, x = points(rng, 40000, 25.0, 1e5)
    with np.errstate(all="ignore"):
        lo = x < a
        for fn, f, side in (("logcdf", sp.gammainc, lo), ("logsf", sp.gammaincc, ~lo)):
            got, direct = getattr(gamma, fn)(x, a, 1.0), f(a, x)
            m = side & (direct > 1e-300)
            assert m.sum() > 10000
            ok = H.close(got[m], np.log(direct[m]), 1e-12)
            assert np.all(ok), (fn, a[m][~ok][:4], x[m][~ok][:4], got[m][~ok][:4], np.log(direct[m][~ok][:4]))
    a, x = points(rng, 1200, 1e5, 1e14)
    lp, lq = gamma.logcdf(x, a, 1.0), gamma.logsf(x, a, 1.0)
    mp.mp.dps = 60
    _, exact = gen.coefficients()
    checked = 0
    for ai, xi, pi, qi in zip(a, x, lp, lq):
        rp, rq = gen.temme_mp(exact, ai, xi)
        ref, got = (rp, pi) if rp is not None else (rq, qi)
        if ref > 1e-290:
            lr = float(mp.log(ref))
            assert abs(got - lr) <= 1e-12 * max(1.0, abs(lr)), (ai, xi, got, lr)
            checked += 1
    assert checked > 600
    pr = rng.uniform(1e-6, 1 - 1e-6, 5000)
    ai = np.exp(rng.uniform(np.log(25.0), np.log(1e5), 5000))
    np.testing.assert_allclose(gamma.ppf(pr, ai, 1.0), sp.gammaincinv(ai, pr), rtol=1e-12)
    ai = np.exp(rng.uniform(np.log(1e5), np.log(1e12), 5000))
    root = gamma.ppf(pr, ai, 1.0)
    back = gamma.cdf(root, ai, 1.0)
    assert np.all(np.abs(back - pr) <= 1e-12 * pr + 2e-14 * np.sqrt(ai / (2 * np.pi)))
    # poisson.logcdf(k, mu) = log Q(k + 1, mu)
    mu = np.exp(rng.uniform(np.log(30.0), np.log(1e9), 400))
    k = np.floor(mu + rng.normal(0, 2.0, mu.size) * np.sqrt(mu))
    got = poisson.logcdf(k, mu)
    for ki, mi, gi in zip(k, mu, got):
        rp, rq = gen.temme_mp(exact, ki + 1.0, mi)
        ref = rq if rq is not None else 1 - rp
        if ref > 1e-290 and ref < 1 - 1e-9:
            lr = float(mp.log(ref))
            assert abs(gi - lr) <= 1e-12 * max(1.0, abs(lr)), (ki, mi, gi, lr)


def test_device_gammaincinv_regimes(gpu):
    from distributions_amd import gamma

    rng = np.random.default_rng(2)
    a = np.exp(rng.uniform(np.log(1e-2), np.log(1e4), 20000))
    p = rng.uniform(0, 1, a.size)
    p[:2000] = np.exp(rng.uniform(-300, 0, 2000))
    p[2000:3000] = 1 - np.exp(rng.uniform(-36, 0, 1000))
    got, exp = gamma.ppf(p, a, 1.0), sp.gammaincinv(a, p)
    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-300)  # subnormal roots carry fewer bits


def test_device_betainc_and_inverse_regimes(gpu):
    from distributions_amd import beta

    rng = np.random.default_rng(3)
    n = 20000
    a = np.exp(rng.uniform(np.log(1e-2), np.log(500), n))
    b = np.exp(rng.uniform(np.log(1e-2), np.log(500), n))
    x = np.clip(rng.beta(np.minimum(a, 50) + 0.1, np.minimum(b, 50) + 0.1), 1e-300, 1 - 1e-16)
    with np.errstate(all="ignore"):
        got, exp = beta.logcdf(x, a, b), np.log(sp.betainc(a, b, x))
        assert np.all(H.close(got, exp))
        got, exp = beta.logsf(x, a, b), np.log(sp.betainc(b, a, 1 - x))
        assert np.all(H.close(got, exp))
    p = rng.uniform(1e-6, 1 - 1e-6, n)
    got, exp = beta.ppf(p, a, b), sp.betaincinv(a, b, p)
    # SciPy's inverse clamps / fails for tiny roots, so the check is the forward equation (as in the host test): the
    # residual may be the forward function's noise plus what 4 eps on the root's small side move it through dI/dx
    with np.errstate(all="ignore"):
        slope = np.exp((a - 1) * np.log(got) + (b - 1) * np.log1p(-got) - sp.betaln(a, b))
    allowed = 1e-12 * np.minimum(p, 1 - p) + slope * (9e-16 * np.minimum(got, 1 - got) + 2.3e-16)
    resid = np.abs(sp.betainc(a, b, got) - p)
    assert np.all(resid <= allowed), (a[resid > allowed][:5], b[resid > allowed][:5], p[resid > allowed][:5])
    both = (exp > 1e-200) & (exp < 1 - 1e-12) & (np.abs(sp.betainc(a, b, exp) - p) <= allowed)
    assert both.mean() > 0.9
    np.testing.assert_allclose(got[both], exp[both], rtol=1e-9)


# --- edge-point values of the literal float64 restatement, as literals ----------------------
def _edge_ok(got, lit):
    got = np.asarray(got, dtype=np.float64)
    lit = np.asarray(lit, dtype=np.float64)
    same = (got == lit) | (np.isnan(got) & np.isnan(lit))
    with np.errstate(all="ignore"):
        return same | (np.isfinite(lit) & (np.abs(got - lit) <= 5e-6 * np.abs(lit)))  # the table has 6 digits


EDGE_TABLE = [
    # (module, fn, params, points, literal values of Appendix C)
    ("normal", "logpdf", (0.0, 1.0), [-inf, -40, -1, 0, 1, 40, inf], [-inf, -800.919, -1.41894, -0.918939, -1.41894, -800.919, -inf]),
    ("normal", "logcdf", (0.0, 1.0), [-inf, -40, -1, 0, 1, 40, inf], [-inf, -804.608, -1.84102, -0.693147, -0.172754, -0.0, -0.0]),
    ("normal", "logsf", (0.0, 1.0), [-inf, -40, -1, 0, 1, 40, inf], [-0.0, -0.0, -0.172754, -0.693147, -1.84102, -804.608, -inf]),
    ("normal", "ppf", (0.0, 1.0), [-0.1, 0, 1e-300, 0.5, 1 - 1e-16, 1, 1.1], [nan, -inf, -inf, 0, 8.20954, inf, nan]),
    ("gamma", "logpdf", (2.0, 1 / 3), [-1, 0, 1e-300, 1, 6, 1e4, inf], [-inf, -inf, -692.973, -2.53056, -2.40547, -3326.32, nan]),
    ("gamma", "logcdf", (2.0, 1 / 3), [-1, 0, 1e-300, 1, 6, 1e4, inf], [-inf, -inf, -inf, -3.10946, -0.520886, -0.0, -0.0]),
    ("gamma", "logsf", (2.0, 1 / 3), [-1, 0, 1e-300, 1, 6, 1e4, inf], [0, -0.0, -0.0, -0.0456513, -0.901388, -inf, -inf]),
    ("gamma", "ppf", (2.0, 1 / 3), [-0.1, 0, 1e-300, 0.5, 1 - 1e-16, 1, 1.1], [nan, 0, 4.24264e-150, 5.03504, 121.385, inf, nan]),
    ("gamma", "logpdf", (0.5, 2.0), [-1, 0, 1e-300, 1, 6, 1e4, inf], [-inf, inf, 345.162, -2.22579, -13.1217, -20004.8, -inf]),
    ("gamma", "logcdf", (0.5, 2.0), [-1, 0, 1e-300, 1, 6, 1e4, inf], [-inf, -inf, -344.92, -0.0465679, -9.63357e-07, -0.0, -0.0]),
    ("gamma", "logsf", (0.5, 2.0), [-1, 0, 1e-300, 1, 6, 1e4, inf], [0, -0.0, -1.59577e-150, -3.09004, -13.8528, -inf, -inf]),
    ("beta", "logpdf", (2.0, 5.0), [-1, 0, 1e-300, 0.5, 1 - 1e-16, 1, 2], [-inf, -inf, -687.374, -0.0645385, -143.546, -inf, -inf]),
    ("beta", "logcdf", (2.0, 5.0), [-1, 0, 1e-300, 0.5, 1 - 1e-16, 1, 2], [-inf, -inf, -inf, -0.115832, 0, 0, 0]),
    ("beta", "logsf", (2.0, 5.0), [-1, 0, 1e-300, 0.5, 1 - 1e-16, 1, 2], [0, 0, 0, -2.21297, -181.892, -inf, -inf]),
    # q = 1e-300: SciPy 1.15.3's betaincinv returns NaN there ("version-dependent, unpinned" in Appendix C): not listed
    ("beta", "ppf", (2.0, 5.0), [-0.1, 0, 0.5, 1 - 1e-16, 1, 1.1], [nan, 0, 0.26445, 0.99955, 1, nan]),
    ("beta", "logpdf", (0.5, 3.0), [0, 1], [inf, -inf]),
    ("beta", "logpdf", (1.0, 1.0), [0, 1], [0, 0]),
    ("studentt", "logpdf", (5.0, 0.0, 2.0), [-inf, -1e10, -1, 0, 1, 1e10, inf], [-inf, -130.83, -1.80814, -1.66177, -1.80814, -130.83, -inf]),
    ("studentt", "logcdf", (5.0, 0.0, 2.0), [-inf, -1e10, -1, 0, 1, 1e10, inf], [-inf, -109.413, -1.1421, -0.693147, -0.384412, -3.03685e-48, -0.0]),
    ("studentt", "logsf", (5.0, 0.0, 2.0), [-inf, -1e10, -1, 0, 1, 1e10, inf], [-0.0, -3.03685e-48, -0.384412, -0.693147, -1.1421, -109.413, -inf]),
    ("studentt", "ppf", (5.0, 0.0, 2.0), [-0.1, 0, 1e-300, 0.5, 1 - 1e-16, 1, 1.1], [nan, -inf, -3.13679e60, 0, 4868.58, inf, nan]),
    ("poisson", "logpdf", (2.0,), [-2, -1, -0.5, 0, 1, 2.5, inf], [-inf, -inf, -2.91894, -2, -1.30685, -1.46811, nan]),
    ("binomial", "logpdf", (10, 0.3), [-1, 0, 3, 3.5, 10, 11, inf], [-inf, -3.56675, -1.32115, -1.41598, -12.0397, -inf, -inf]),
]


@pytest.mark.parametrize("name, fn, params, pts, lit", EDGE_TABLE, ids=[f"{e[0]}.{e[1]}{e[2]}" for e in EDGE_TABLE])
def test_appendix_c_edge_table(gpu, name, fn, params, pts, lit):
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    pts = np.asarray(pts, dtype=np.float64)
    got = getattr(m, fn)(pts, *params)
    ok = _edge_ok(got, lit)
    assert np.all(ok), (name, fn, pts[~ok], got[~ok], np.asarray(lit)[~ok])
    # and the full-precision oracle at the same points, no masks
    with np.errstate(all="ignore"):
        exp = R.TABLE[f"{name}_{fn}"](pts, *params)
    ok = H.close(got, exp)
    assert np.all(ok), (name, fn, pts[~ok], got[~ok], exp[~ok])


def test_core_deep_tails_log_scale(gpu):
    """logcdf / logsf of the core four where the probability underflows a probability-scale comparison:
    the log values themselves must agree at 1e-12 (no cancellation mask)."""
    from distributions_amd import beta, gamma, normal, studentt

    with np.errstate(all="ignore"):
        z = np.concatenate([-np.linspace(5, 38, 400), -np.exp(np.linspace(np.log(38), np.log(1e150), 200))])
        for fn, pts in (("logcdf", z), ("logsf", -z)):
            got, exp = getattr(normal, fn)(pts, 0.0, 1.0), R.TABLE[f"normal_{fn}"](pts, 0.0, 1.0)
            assert np.all(H.close(got, exp)), fn
            assert np.all(got[np.abs(pts) < 1e150] < -13)  # really the deep tail
        x = np.exp(np.linspace(np.log(40), np.log(700), 300))
        got, exp = gamma.logsf(x, 2.0, 1.0), R.gamma_logsf(x, 2.0, 1.0)
        assert np.all(H.close(got, exp)) and got[-1] < -600
        x = np.exp(np.linspace(np.log(1e-150), np.log(1e-3), 300))
        got, exp = gamma.logcdf(x, 2.5, 1.3), R.gamma_logcdf(x, 2.5, 1.3)
        assert np.all(H.close(got, exp)) and got[0] < -600
        got, exp = beta.logcdf(x, 2.0, 5.0), R.beta_logcdf(x, 2.0, 5.0)
        assert np.all(H.close(got, exp)) and got[0] < -600
        got, exp = beta.logsf(1 - x[x > 1e-15], 2.0, 5.0), R.beta_logsf(1 - x[x > 1e-15], 2.0, 5.0)
        assert np.all(H.close(got, exp))
        t = -np.exp(np.linspace(np.log(10), np.log(1e60), 300))
        got, exp = studentt.logcdf(t, 5.0, 0.0, 2.0), R.studentt_logcdf(t, 5.0, 0.0, 2.0)
        assert np.all(H.close(got, exp)) and got[-1] < -600
        got, exp = studentt.logsf(-t, 5.0, 0.0, 2.0), R.studentt_logsf(-t, 5.0, 0.0, 2.0)
        assert np.all(H.close(got, exp))


def test_studentt_large_nu_arm(gpu):
    """Quirk Q1 (studentt.py:6-8, nu > 1e5): logpdf is normal.logcdf(x, mu, sigma), logcdf is
    normal.cdf(z, mu, sigma) of the already standardised z -- reproduced bug-for-bug; parity unpinned by the
    reference (its tests stop at nu = 100), pinned by the restatement."""
    from distributions_amd import normal, studentt

    rng = np.random.default_rng(21)
    for nu, mu, sigma in ((2e5, 0.0, 1.0), (2e5, 1.5, 2.0), (1e7, -3.0, 0.25)):
        x = np.concatenate([mu + sigma * rng.standard_normal(5000) * 3, [mu, -inf, inf, mu - 30 * sigma, mu + 30 * sigma]])
        with np.errstate(all="ignore"):
            for fn in ("logpdf", "logcdf", "logsf", "pdf", "cdf", "sf"):
                got, exp = getattr(studentt, fn)(x, nu, mu, sigma), R.TABLE[f"studentt_{fn}"](x, nu, mu, sigma)
                ok = H.close(got, exp)
                assert np.all(ok), (nu, fn, x[~ok][:4], got[~ok][:4], exp[~ok][:4])
        # the arm really is the mis-imported one, and the boundary nu = 1e5 itself is on the exact-formula side
        # (normal's own scalar-parameter kernel reads log Phi from its LDS table, the arm inside studentt's kernel
        # evaluates the erfcx form: the same function to 1e-15, no longer the same bits)
        a, b = studentt.logpdf(x, nu, mu, sigma), normal.logcdf(x, mu, sigma)
        assert np.all((a == b) | (np.abs(a - b) <= 2e-15 * np.maximum(1.0, np.abs(b))))
        assert np.array_equal(studentt.logcdf(x, nu, mu, sigma), normal.cdf((x - mu) / sigma, mu, sigma))
    # nu = 1e5 itself is on the exact-formula side.  There gammaln((nu+1)/2) - gammaln(nu/2) subtracts two values of
    # 4.9e5: one ulp of either is 5.8e-11 on the result in the reference itself (measured against mpmath: SciPy's
    # gammaln chain 3.8e-11, ours 7.9e-11), so the bar is 1e-12 + 4 ulp of gammaln(nu/2)
    x = rng.standard_normal(1000)
    got, exp = studentt.logpdf(x, 1e5, 0.0, 1.0), R.studentt_logpdf(x, 1e5, 0.0, 1.0)
    assert np.all(np.abs(got - exp) <= 1e-12 * np.maximum(1, np.abs(exp)) + 4 * np.spacing(sp.gammaln(5e4)))
    assert not np.array_equal(got, normal.logcdf(x, 0.0, 1.0))
    # nu as an array straddling the switch (k_generic: derive per element)
    nus = np.where(np.arange(1000) % 2 == 0, 50.0, 3e5)  # (nu = 50: the gammaln difference is exact to 1e-14)
    for fn in ("logpdf", "logcdf"):
        assert np.all(H.close(getattr(studentt, fn)(x, nus, 0.2, 1.1), R.TABLE[f"studentt_{fn}"](x, nus, 0.2, 1.1)))


def test_config1_as_specified(gpu):
    """BASELINE.json configs[0] / SURVEY.md 8(d) config 1: N = 1e6 fp64, x = default_rng(0).normal(0.3, 5.1, N),
    mu = 0.3, sigma = 1.7 -- |z| reaches ~15, i.e. the far tail of the erfcx arm of normal.logcdf."""
    from distributions_amd import normal

    x = np.random.default_rng(0).normal(0.3, 5.1, 1_000_000)
    assert np.abs((x - 0.3) / 1.7).max() > 13
    for fn in ("logpdf", "logcdf", "logsf"):
        got, exp = getattr(normal, fn)(x, 0.3, 1.7), R.TABLE[f"normal_{fn}"](x, 0.3, 1.7)
        ok = H.close(got, exp)
        assert np.all(ok), (fn, x[~ok][:4], got[~ok][:4], exp[~ok][:4])
    got = normal.sum_logpdf(x, 0.3, 1.7)
    terms = R.normal_logpdf(x, 0.3, 1.7)
    assert abs(got - math.fsum(terms)) <= 1e-12 * float(np.abs(terms).sum())
