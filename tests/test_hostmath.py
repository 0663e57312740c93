"""CPU unit tests of the device scalar bodies (csrc/m355_dists.h, m355_special.h, m355_math.h)
compiled for the host with g++ (tests/hostmath/hostmath.cpp).  These check the series, continued
fractions and root finders against the oracle while no GPU is attached.  They are NOT the parity
tests: the hardware seeds and the device libm differ on the GPU; parity proper is `-m gpu`."""

import ctypes
import sys
import os
import subprocess
import types

import numpy as np
import pytest
from numpy.testing import assert_allclose
from scipy import special as sp

from oracle import restatement as R
from tests import harness as H

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostmath", "hostmath.cpp")
LIB = os.path.join(HERE, "hostmath", "libhostmath.so")
from distributions_amd._lib import DIST, FN  # ids of include/m355_dist.h (pure Python, no library load)
_dp = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="session")
def hostlib():
    csrc = os.path.join(HERE, "..", "distributions_amd", "csrc")
    deps = [SRC] + [os.path.join(csrc, f) for f in ("m355_math.h", "m355_special.h", "m355_dists.h", "m355_dists2.h", "m355_dists3.h", "m355_dists4.h", "m355_grad.h",
                                                    "m355_dist_list.h", "m355_rng.h", "m355_zigtab.h", "m355_moments.h", "m355_hermgauss70.h", "m355_logfact128.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        tmp = f"{LIB}.{os.getpid()}.tmp"  # (several xdist workers may build at once: each writes its own file, the rename is atomic)
        subprocess.check_call(["g++", "-O2", "-mfma", "-std=c++17", "-shared", "-fPIC", "-o", tmp, SRC])
        os.replace(tmp, LIB)
    return ctypes.CDLL(LIB)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def host_namespace(lib, name, fns):
    ns = types.SimpleNamespace()

    def make(fn):
        def call(x, *params):
            x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
            p = np.ascontiguousarray(np.asarray(params, dtype=np.float64))
            out = np.empty_like(x)
            rc = lib.hm_eval(DIST[name], FN[fn], ctypes.c_long(x.size), _ptr(x), _ptr(p), _ptr(out))
            assert rc == 0
            return out

        return call

    for fn in fns:
        setattr(ns, fn, make(fn))
    return ns


def vec(lib, name, *arrs):
    arrs = np.broadcast_arrays(*[np.asarray(a, dtype=np.float64) for a in arrs])
    arrs = [np.ascontiguousarray(a).ravel() for a in arrs]
    out = np.empty_like(arrs[0])
    getattr(lib, name)(ctypes.c_long(out.size), *[_ptr(a) for a in arrs], _ptr(out))
    return out


HOST_FNS = {
    # erfcinv exists only in the device libm (ppf / isf of truncatednormal: GPU-tested); erfcx / erfinv are the
    # hand-written ones of m355_special.h and run here too
    "truncatednormal": ["logpdf", "pdf", "cdf", "logcdf", "sf", "logsf"],
}
for _name in DIST:
    HOST_FNS.setdefault(_name, list(FN))


@pytest.mark.parametrize("name, sp_dist, params, sp_params, support, kw", H.grids(), ids=H.grid_ids())
def test_host_bodies_harness_a(hostlib, name, sp_dist, params, sp_params, support, kw):
    H.run_distribution_checks(host_namespace(hostlib, name, HOST_FNS[name]), sp_dist, params,
                              sp_params, support, **kw)


def _close(got, exp, tol=1e-12):
    """The parity criterion of BASELINE.md: |a-b| <= tol*max(1,|b|), NaN==NaN, inf==inf."""
    got, exp = np.asarray(got), np.asarray(exp)
    same = (got == exp) | (np.isnan(got) & np.isnan(exp))
    with np.errstate(all="ignore"):
        ok = same | (np.abs(got - exp) <= tol * np.maximum(1.0, np.abs(exp)))
    return ok


def test_host_log_log1p(hostlib):
    rng = np.random.default_rng(0)
    x = np.concatenate([np.exp(rng.uniform(-740, 700, 100000)), rng.uniform(0.5, 2, 100000),
                        [1.0, 0.0, np.inf, -1.0, 5e-324, 1e-310, np.nan]])
    with np.errstate(all="ignore"):
        exp = np.log(x)
    got = vec(hostlib, "hm_log", x)
    assert np.all(_close(got, exp, 1e-15) | (np.abs(got - exp) <= 4e-16 * np.abs(exp)))
    w = np.concatenate([rng.uniform(-1, 1, 100000), np.exp(rng.uniform(-50, 50, 50000)),
                        -np.exp(rng.uniform(-50, 0, 50000)), [0, -1, np.inf, -2, np.nan]])
    with np.errstate(all="ignore"):
        exp = np.log1p(w)
    got = vec(hostlib, "hm_log1p", w)
    with np.errstate(all="ignore"):
        assert np.all(_close(got, exp, 1e-15) | (np.abs(got - exp) <= 6e-16 * np.abs(exp)))


def test_host_erfcx_erfinv(hostlib):
    """The hand-written erfcx (one polynomial in (a-4)/(a+4) on the half line) and erfinv (Giles' split in
    w = -log(1-x^2)) against SciPy: relative error far inside 1e-12, special values exact."""
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.uniform(-5, 30, 200000), np.exp(rng.uniform(-40, 40, 50000)), -np.exp(rng.uniform(-40, 3.2, 50000)),
                        [0.0, -0.0, np.inf, -26.0, -27.0, 1e300, np.nan]])
    with np.errstate(all="ignore"):
        got, exp = vec(hostlib, "hm_erfcx", x), sp.erfcx(x)
    fin = np.isfinite(exp)
    # x < 0: SciPy's 2 exp(x^2) - erfcx(-x) rounds x^2 before the exponential (x^2 * 1.1e-16 relative; ours splits
    # x^2 exactly and stays at 1e-16 against mpmath), hence the x^2 term of the allowance
    # (design accuracy of erfcx_pos: 2.8e-15, the degree-20 polynomial)
    tol = 1e-14 + np.where(x < 0, 2.3e-16 * x * x, 0.0)
    assert np.all(np.abs(got[fin] - exp[fin]) <= tol[fin] * np.abs(exp[fin]))
    assert np.array_equal(np.isnan(got), np.isnan(exp)) and np.array_equal(np.isposinf(got), np.isposinf(exp))
    assert got[x == np.inf][0] == 0.0
    q = np.concatenate([rng.uniform(-1, 1, 200000), 1 - np.exp(rng.uniform(-36.7, 0, 50000)), np.exp(rng.uniform(-36.7, 0, 50000)) - 1,
                        np.exp(rng.uniform(-300, 0, 20000)), [0.0, 1.0, -1.0, 1 - 2.0**-53, 2.0**-53 - 1, 1.5, -2.0, np.nan, 5e-324]])
    with np.errstate(all="ignore"):
        got, exp = vec(hostlib, "hm_erfinv", q), sp.erfinv(q)
    fin = np.isfinite(exp)
    assert np.all(np.abs(got[fin] - exp[fin]) <= 4e-15 * np.abs(exp[fin])), np.max(np.abs(got[fin] - exp[fin]) / np.abs(exp[fin]))
    assert np.array_equal(got[~fin & ~np.isnan(exp)], exp[~fin & ~np.isnan(exp)])  # +-inf at +-1
    assert np.array_equal(np.isnan(got), np.isnan(exp))  # |x| > 1 and NaN


def test_host_logphi_table(hostlib):
    """normal_logcdf_z_tab (piecewise degree-8 polynomials on [-10, 6), the exact body outside): against mpmath at
    4e-16 max(1, |log Phi|) on the table's range, cell boundaries and both ends included, and against SciPy's log_ndtr
    at 1e-13 everywhere (NaN, infinities and the far tails go through normal_logcdf_z)."""
    import mpmath as mp
    from scipy.special import log_ndtr

    rng = np.random.default_rng(17)
    edges = -10.0 + np.arange(0, 257) / 16.0
    z = np.concatenate([rng.uniform(-10, 6, 4000), edges, np.nextafter(edges, -np.inf), np.nextafter(edges, np.inf)])
    got = vec(hostlib, "hm_logphi_tab", z)
    mp.mp.dps = 40
    for zi, gi in zip(z, got):
        ref = mp.log(mp.ncdf(mp.mpf(zi)))
        assert abs(gi - ref) <= 4e-16 * max(1, abs(ref)), (zi, gi, float(ref))
    z = np.concatenate([rng.normal(0, 6, 200000), [-40.0, -10.0, 6.0, 38.0, np.inf, -np.inf, np.nan, 0.0, -0.0]])
    got, exp = vec(hostlib, "hm_logphi_tab", z), log_ndtr(z)
    ok = (got == exp) | (np.isnan(got) & np.isnan(exp)) | (np.abs(got - exp) <= 1e-13 * np.maximum(1.0, np.abs(exp)))
    assert np.all(ok), (z[~ok][:5], got[~ok][:5], exp[~ok][:5])


def test_host_sincos_2pi_and_u53(hostlib):
    """The Box-Muller angle (two degree-7 polynomials after an exact reduction of 4u) against numpy at 1e-15, exact at
    the quadrant points; u53 stays strictly inside (0, 1) at both ends of its 52-bit range."""
    rng = np.random.default_rng(9)
    u = np.concatenate([rng.uniform(0, 1, 200000), [0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 2.0**-53, 1 - 2.0**-53]])
    out = np.empty(2 * u.size)
    hostlib.hm_sincos2pi(ctypes.c_long(u.size), _ptr(u), _ptr(out))
    s, c = out[0::2], out[1::2]
    assert np.max(np.abs(s - np.sin(2 * np.pi * u))) <= 1e-15 and np.max(np.abs(c - np.cos(2 * np.pi * u))) <= 1e-15
    assert np.max(np.abs(s * s + c * c - 1)) <= 1e-15
    for uu, ss, cc in ((0.0, 0.0, 1.0), (0.25, 1.0, 0.0), (0.5, 0.0, -1.0), (0.75, -1.0, 0.0)):
        i = int(np.where(u == uu)[0][-1])
        assert s[i] == ss and c[i] == cc
    hostlib.hm_u53.restype = ctypes.c_double
    hostlib.hm_u53.argtypes = [ctypes.c_uint, ctypes.c_uint]
    assert hostlib.hm_u53(0xFFFFFFFF, 0xFFFFFFFF) == 1 - 2.0**-53 and hostlib.hm_u53(0, 0) == 2.0**-53


def test_host_streaming_logs(hostlib):
    """flog_stream / flog1p_stream (one Newton step on a 24-bit reciprocal seed): <= 1e-14."""
    rng = np.random.default_rng(5)
    x = np.concatenate([np.exp(rng.uniform(-740, 700, 100000)), rng.uniform(0.5, 2, 100000),
                        1 + rng.uniform(-1e-9, 1e-9, 1000), [1.0, 0.0, np.inf, -1.0, 5e-324, 1e-310, np.nan]])
    with np.errstate(all="ignore"):
        exp = np.log(x)
        got = vec(hostlib, "hm_log_stream", x)
        assert np.all(_close(got, exp, 1e-15) | (np.abs(got - exp) <= 1e-14 * np.abs(exp)))
        w = np.concatenate([np.exp(rng.uniform(-60, 60, 100000)), rng.uniform(0, 1, 100000), [0.0, np.inf, np.nan]])
        exp = np.log1p(w)
        got = vec(hostlib, "hm_log1p_stream", w)
        assert np.all(_close(got, exp, 1e-15) | (np.abs(got - exp) <= 1e-14 * np.abs(exp)))


@pytest.mark.parametrize("arange", [(1e-3, 1), (1, 10), (10, 100), (100, 1e4)])
def test_host_gammainc(hostlib, arange):
    rng = np.random.default_rng(1)
    a = np.exp(rng.uniform(np.log(arange[0]), np.log(arange[1]), 20000))
    x = a * np.exp(rng.normal(0, 1.0, a.size))
    for name, f in (("hm_gammainc", sp.gammainc), ("hm_gammaincc", sp.gammaincc)):
        got, exp = vec(hostlib, name, a, x), f(a, x)
        # SciPy's own igam loses ~1e-11 for a > 100 far from the mean (checked against mpmath
        # below, where these bodies hold 1e-12), so the SciPy comparison is looser there.
        tol = 1e-12 if arange[1] <= 100 else 1e-10
        m = exp > 1e-300  # subnormal results carry fewer bits
        with np.errstate(all="ignore"):
            assert np.all(_close(np.log(got[m]), np.log(exp[m]), tol)), name


def _temme_points(rng, n, lo, hi):
    a = np.exp(rng.uniform(np.log(lo), np.log(hi), n))
    sig = np.where(rng.uniform(size=n) < 0.5, rng.uniform(-0.3, 0.3, n), rng.normal(0, 2.0, n) / np.sqrt(a))
    return a, a * (1 + np.clip(sig, -0.3, 0.3))


def test_host_gammainc_temme_region(hostlib):
    """a >= 25, |x - a| <= 0.3 a (Temme's uniform expansion, gammainc_temme), including |x - a| ~ sqrt(a), where P and Q
    are both O(1) and the series / continued fraction would need ~sqrt(a) terms.  exp(-a phi) carries a phi ulps, hence
    the comparison of logs at 1e-12 max(1, |log|).
      a <= 1e5: against SciPy (igam / igamc take the same expansion there and hold 7e-16 in this measure);
      a <= 1e14: against the expansion evaluated with exact rational coefficients in 60-digit arithmetic
                 (tools/gen_temme_coeffs.py: agrees with mpmath's gammainc to 4e-25 for 200 <= a <= 1e4, and with direct
                 quadrature of the density at a = 7.2e12, 2e9, 3.3e7).  SciPy is NOT the reference there: its tails
                 drift from 5e-11 (a ~ 1e5) to tens of percent (a >= 1e8; e.g. gammainc(7207594366627.789,
                 7207581856025.457) = 5.7e-9 against 1.58126e-6 by quadrature), so at such a the library follows the
                 function, not the oracle's error -- DESIGN.md section 4."""
    import mpmath as mp

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import gen_temme_coeffs as gen

    rng = np.random.default_rng(21)
    a, x = _temme_points(rng, 40000, 25.0, 1e5)
    a = np.concatenate([a, [25.0, 25.0, 25.0]])
    x = np.concatenate([x, [25.0 * 0.7, 25.0, 25.0 * 1.3]])
    for name, f in (("hm_gammainc", sp.gammainc), ("hm_gammaincc", sp.gammaincc)):
        got, exp = vec(hostlib, name, a, x), f(a, x)
        m = exp > 1e-300
        with np.errstate(all="ignore"):
            assert np.all(_close(np.log(got[m]), np.log(exp[m]), 1e-12)), name
        assert np.all(got[~m] <= 1e-299)
    a, x = _temme_points(rng, 1200, 1e5, 1e14)
    a = np.concatenate([a, [1e300, 4e15]])
    x = np.concatenate([x, [1e300, 4e15 + 1]])
    p, q = vec(hostlib, "hm_gammainc", a, x), vec(hostlib, "hm_gammaincc", a, x)
    assert np.max(np.abs(p + q - 1)) <= 4e-16
    mp.mp.dps = 60
    _, exact = gen.coefficients()
    for ai, xi, pi, qi in zip(a[:-2], x[:-2], p, q):
        rp, rq = gen.temme_mp(exact, ai, xi)
        ref, got = (rp, pi) if rp is not None else (rq, qi)
        if ref > 1e-290:
            lr = float(mp.log(ref))
            assert abs(np.log(got) - lr) <= 1e-12 * max(1.0, abs(lr)), (ai, xi, got, float(ref))
    # a = 1e300: eta = 0, P = 1/2 - c_0(0)/sqrt(2 pi a)
    assert abs(p[-2] - 0.5) <= 2.3e-16 and abs(q[-2] - 0.5) <= 2.3e-16
    # the quantile through the same region: against SciPy where it is accurate, by the forward function beyond
    pr = rng.uniform(1e-6, 1 - 1e-6, 5000)
    ai = np.exp(rng.uniform(np.log(25.0), np.log(1e5), 5000))
    assert_allclose(vec(hostlib, "hm_gammaincinv", ai, pr), sp.gammaincinv(ai, pr), rtol=1e-12)
    ai = np.exp(rng.uniform(np.log(1e5), np.log(1e12), 5000))
    root = vec(hostlib, "hm_gammaincinv", ai, pr)
    back = vec(hostlib, "hm_gammainc", ai, root)
    # dP/dx <= 1/sqrt(2 pi a): a relative error r of the root moves P by r a / sqrt(2 pi a); r <= 2e-14 (bar: 1e-12)
    assert np.all(np.abs(back - pr) <= 1e-12 * pr + 2e-14 * np.sqrt(ai / (2 * np.pi)))


def test_host_gammainc_betainc_vs_mpmath(hostlib):
    import mpmath as mp

    mp.mp.dps = 40
    rng = np.random.default_rng(11)
    a = np.exp(rng.uniform(np.log(0.05), np.log(5e3), 60))
    x = a * np.exp(rng.normal(0, 0.7, a.size))
    gp, gq = vec(hostlib, "hm_gammainc", a, x), vec(hostlib, "hm_gammaincc", a, x)
    for ai, xi, pi, qi in zip(a, x, gp, gq):
        tp = mp.gammainc(ai, 0, xi, regularized=True)
        tq = mp.gammainc(ai, xi, mp.inf, regularized=True)
        if tp > 1e-290:
            assert abs(mp.log(pi) - mp.log(tp)) <= 1e-12 * max(1, abs(mp.log(tp))), (ai, xi)
        if tq > 1e-290:
            assert abs(mp.log(qi) - mp.log(tq)) <= 1e-12 * max(1, abs(mp.log(tq))), (ai, xi)
    a = np.exp(rng.uniform(np.log(0.05), np.log(300), 40))
    b = np.exp(rng.uniform(np.log(0.05), np.log(300), 40))
    xb = np.clip(rng.beta(np.minimum(a, 30) + 0.1, np.minimum(b, 30) + 0.1), 1e-12, 1 - 1e-12)
    gi = vec(hostlib, "hm_betainc", a, b, xb)
    for ai, bi, xi, ii in zip(a, b, xb, gi):
        try:
            t = mp.betainc(ai, bi, 0, xi, regularized=True)
        except Exception:
            continue  # mpmath's hypergeometric series fails to converge for some large a, b
        if t > 1e-290:
            assert abs(mp.log(ii) - mp.log(t)) <= 1e-12 * max(1, abs(mp.log(t))), (ai, bi, xi)


def test_host_gammaincinv(hostlib):
    rng = np.random.default_rng(2)
    a = np.exp(rng.uniform(np.log(1e-2), np.log(1e4), 20000))
    p = rng.uniform(0, 1, a.size)
    p[:2000] = np.exp(rng.uniform(-300, 0, 2000))
    p[2000:3000] = 1 - np.exp(rng.uniform(-36, 0, 1000))
    got, exp = vec(hostlib, "hm_gammaincinv", a, p), sp.gammaincinv(a, p)
    assert_allclose(got, exp, rtol=1e-12, atol=1e-300)  # subnormal roots carry fewer bits


def test_host_betainc_and_inverse(hostlib):
    rng = np.random.default_rng(3)
    n = 20000
    a = np.exp(rng.uniform(np.log(1e-2), np.log(500), n))
    b = np.exp(rng.uniform(np.log(1e-2), np.log(500), n))
    x = np.clip(rng.beta(np.minimum(a, 50) + 0.1, np.minimum(b, 50) + 0.1), 1e-300, 1 - 1e-16)
    got, exp = vec(hostlib, "hm_betainc", a, b, x), sp.betainc(a, b, x)
    with np.errstate(all="ignore"):
        assert np.all(_close(np.log(got), np.log(exp)))
    p = rng.uniform(1e-6, 1 - 1e-6, n)
    got, exp = vec(hostlib, "hm_betaincinv", a, b, p), sp.betaincinv(a, b, p)
    # SciPy's inverse clamps / fails for tiny roots, so the check is the forward equation itself:
    # the residual may be the forward function's noise plus what 4 eps on the root's small side
    # (and one rounding of x = 1 - y) move it through the slope dI/dx.
    with np.errstate(all="ignore"):
        slope = np.exp((a - 1) * np.log(got) + (b - 1) * np.log1p(-got) - sp.betaln(a, b))
    allowed = 1e-12 * np.minimum(p, 1 - p) + slope * (9e-16 * np.minimum(got, 1 - got) + 2.3e-16)
    resid = np.abs(sp.betainc(a, b, got) - p)
    assert np.all(resid <= allowed), (a[resid > allowed][:5], b[resid > allowed][:5], p[resid > allowed][:5])
    both = (exp > 1e-200) & (exp < 1 - 1e-12) & (np.abs(sp.betainc(a, b, exp) - p) <= allowed)
    assert both.mean() > 0.9
    assert_allclose(got[both], exp[both], rtol=1e-9)


@pytest.mark.parametrize("name, sp_dist, params, sp_params, support, kw", H.grids(), ids=H.grid_ids())
def test_host_bodies_vs_oracle_1e12(hostlib, name, sp_dist, params, sp_params, support, kw):
    """Tight (1e-12) comparison with the oracle on a denser point set than harness A."""
    rng = np.random.default_rng(7)
    d = sp_dist(**sp_params)
    xs = np.concatenate([np.asarray(d.rvs(2000, random_state=rng), dtype=np.float64),
                         H.extended_vals(d, support)])
    ns = host_namespace(hostlib, name, HOST_FNS[name])
    nx, nq = H.SLOW_ORACLE.get(name, (len(xs), 2000))
    for fn in HOST_FNS[name]:
        if fn in ("ppf", "isf"):
            pts = np.concatenate([rng.uniform(0, 1, nq), H.ISF_POINTS, [1e-300, 1e-12, 1 - 1e-12]])
        else:
            pts = xs[-nx:]
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"{name}_{fn}"](pts, *params)
        got = getattr(ns, fn)(pts, *params)
        ok = _close(got, exp) | H.family_ok(name, fn, got, exp, params) | H.ppf_unpinned(name, fn, pts)
        if fn in ("ppf", "isf") and name in ("beta", "studentt", "halfstudentt", "betascaled", "skew_studentt"):
            # the oracle's betaincinv (SciPy) returns NaN / clamps for extreme tails (SURVEY App. C,
            # "version-dependent, unpinned"); studentt / halfstudentt amplify 1 ulp of betaincinv through 1/x - 1
            # (reference formula, studentt.py:85-86): modelled allowance in harness.ppf_close, 1e-12 otherwise
            ok |= np.isnan(exp) | H.ppf_close(name, fn, params, got, exp)
            ok |= (pts < 1e-100) | (pts > 1 - 1e-15)
        assert np.all(ok), (name, fn, params, pts[~ok][:5], got[~ok][:5], exp[~ok][:5])


def _discrete_cases():
    out = [(g[0], list(g[2]), g[4]) for g in H.discrete_grids()]
    out += [(n, p, (0, 60)) for n, p in H.empirical_discrete_cases()]
    return out


@pytest.mark.parametrize("name, params, support", _discrete_cases(),
                         ids=[f"{n}-{'-'.join(map(str, p))}" for n, p, _ in _discrete_cases()])
def test_host_discrete_vs_oracle(hostlib, name, params, support):
    """Discrete families on integer and non-integer points: values at 1e-12,
    ppf / isf (integer results of find_ppf_discrete, optimization.py:102-129) bit-exact."""
    rng = np.random.default_rng(9)
    hi = 60 if not np.isfinite(support[1]) else int(support[1]) + 4
    xs, qs = H.discrete_points(rng, hi=max(hi, 8), n=1500)
    ns = host_namespace(hostlib, name, list(FN))
    for fn in FN:
        pts = qs if fn in ("ppf", "isf") else xs
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"{name}_{fn}"](pts, *params)
        got = getattr(ns, fn)(pts, *params)
        if fn in ("ppf", "isf"):
            ok = (got == exp) | (np.isnan(got) & np.isnan(exp))
        else:
            ok = _close(got, exp) | H.family_ok(name, fn, got, exp, params)
        assert np.all(ok), (name, fn, params, pts[~ok][:5], got[~ok][:5], exp[~ok][:5])


def test_host_digamma(hostlib):
    rng = np.random.default_rng(12)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-6), np.log(1e8), 50000)), -rng.uniform(0.01, 30, 2000),
                        [1.0, 2.0, 0.5, 1.4616321449683623, 10.0, 1e-300, np.inf]])
    got, exp = vec(hostlib, "hm_digamma", x), sp.digamma(x)
    with np.errstate(all="ignore"):
        near_pole = (x < 0) & (np.abs(x - np.round(x)) < 0.05)  # conditioned like 1 / distance to the pole
    assert np.all(_close(got, exp, 1e-14) | near_pole)
    assert np.all(_close(got, exp, 1e-11) | ~near_pole)
    pos = (x > 0) & np.isfinite(x)
    assert np.all(np.abs(got[pos] - exp[pos]) <= 2e-15 * np.maximum(1.0, np.abs(exp[pos])))


def host_grad(lib, name, x, params):
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    p = np.ascontiguousarray(np.asarray(list(params) + [0.0] * 4, dtype=np.float64)[:4])
    out = np.empty((x.size, 2 + len(params)))
    assert lib.hm_grad(DIST[name], ctypes.c_long(x.size), _ptr(x), _ptr(p), _ptr(out)) == 0
    return out


_GRAD_CASES = [g for g in H.grids() if g[0] in __import__("distributions_amd._lib", fromlist=["GRAD"]).GRAD]


@pytest.mark.parametrize("name, sp_dist, params, sp_params, support, kw", _GRAD_CASES,
                         ids=[f"{g[0]}-{'-'.join(str(v) for v in g[2])}" for g in _GRAD_CASES])
def test_host_grad_vs_oracle(hostlib, name, sp_dist, params, sp_params, support, kw):
    """Dual-number bodies of csrc/m355_grad.h against oracle/restatement_grad.py at 1e-12 (value and every
    partial), on draws from the distribution plus the support ends and out-of-support points."""
    from oracle import restatement_grad as G

    if name == "logitnormal" or (name in ("studentt", "halfstudentt") and params[0] > 1e5):
        pytest.skip("erfcx / device-libm-only arm: GPU-tested")
    rng = np.random.default_rng(29)
    d = sp_dist(**sp_params)
    xs = np.concatenate([np.asarray(d.rvs(2000, random_state=rng), dtype=np.float64), H.extended_vals(d, support)])
    xs = xs[np.isfinite(xs)]
    got = host_grad(hostlib, name, xs, params)
    val, grads = G.logpdf_grad(name, xs, *params)
    assert H.grad_close(got.T, [val] + grads, (name, params, xs))


@pytest.mark.parametrize("h, z", H.POLYAGAMMA_CASES)
def test_host_polyagamma_vs_oracle(hostlib, h, z):
    """Series density and trapezoid cdf of polyagamma (csrc/m355_dists4.h) against the oracle at 1e-12; ppf / isf
    start from erfinv (device libm only) and are GPU-tested."""
    xs, _ = H.polyagamma_points(np.random.default_rng(41), h, z, nx=300)
    fns = ["logpdf", "pdf", "cdf", "logcdf", "sf", "logsf"]
    ns = host_namespace(hostlib, "polyagamma", fns)
    for fn in fns:
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"polyagamma_{fn}"](xs, h, z)
        got = getattr(ns, fn)(xs, h, z)
        ok = _close(got, exp) | H.family_ok("polyagamma", fn, got, exp)
        assert np.all(ok), (fn, h, z, xs[~ok][:5], got[~ok][:5], exp[~ok][:5])


def _host_sum_acc(lib, name, x, params):
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    p = np.ascontiguousarray(np.asarray(list(params) + [0.0] * 4, dtype=np.float64)[:4])
    out = ctypes.c_double()
    rc = lib.hm_sum_acc(DIST[name], ctypes.c_long(x.size), _ptr(x), _ptr(p), ctypes.byref(out))
    return rc, out.value


_SUM_ACC_CASES = [
    ("gamma", (2.5, 1.3), lambda r, n: r.gamma(2.5, 1 / 1.3, n)),
    ("gamma", (0.01, 10.0), lambda r, n: np.maximum(r.gamma(0.01, 0.1, n), 1e-300)),
    ("gamma", (1.0, 1.0), lambda r, n: r.gamma(1.0, 1.0, n)),
    ("gamma", (900.0, 3.0), lambda r, n: r.gamma(900.0, 1 / 3.0, n)),
    ("beta", (2.0, 5.0), lambda r, n: r.beta(2, 5, n)),
    ("beta", (0.5, 0.5), lambda r, n: np.clip(r.beta(0.5, 0.5, n), 1e-300, 1 - 1e-16)),
    ("beta", (800.0, 1.0), lambda r, n: r.beta(800, 1, n)),       # x just below 1: log x ~ -1e-3
    ("beta", (1.0, 1.0), lambda r, n: r.uniform(0, 1, n)),
    ("studentt", (5.0, 0.0, 2.0), lambda r, n: 2 * r.standard_t(5, n)),
    ("studentt", (1.0, 3.0, 0.01), lambda r, n: 3 + 0.01 * r.standard_t(1, n)),   # cauchy tails: 1 + w up to 1e30
    ("studentt", (2000.0, 0.0, 1.0), lambda r, n: 1e-9 * r.normal(size=n)),      # w ~ 1e-21: the side sum carries it
]
_SUM_ACC_ONLY = [  # families with the product form on the sum-logp path only (no sufficient-statistics gradient)
    ("cauchy", (0.5, 2.0), lambda r, n: 0.5 + 2.0 * r.standard_cauchy(n)),
    ("halfcauchy", (3.5,), lambda r, n: np.abs(3.5 * r.standard_cauchy(n))),
    ("pareto", (2.5, 1.5), lambda r, n: 1.5 * (1 + r.pareto(2.5, n))),
    ("inversegamma", (3.0, 2.0), lambda r, n: 2.0 / r.gamma(3.0, 1.0, n)),
    ("chisquared", (7.0,), lambda r, n: r.chisquare(7.0, n)),
    ("chisquared", (2.0,), lambda r, n: r.chisquare(2.0, n)),
    ("scaledinversechisquared", (5.0, 0.7), lambda r, n: 5.0 * 0.7 / r.chisquare(5.0, n)),
    ("betascaled", (2.0, 5.0, -1.0, 3.0), lambda r, n: -1.0 + 4.0 * r.beta(2, 5, n)),
]


@pytest.mark.parametrize("name, params, draw", _SUM_ACC_CASES + _SUM_ACC_ONLY, ids=[f"{c[0]}-{c[1]}" for c in _SUM_ACC_CASES + _SUM_ACC_ONLY])
def test_host_sum_acc_vs_fsum(hostlib, name, params, draw):
    """Dist<>::SumAcc (sum of logpdf with sums of logarithms carried as products, csrc/m355_math.h LogProd)
    against math.fsum of the oracle's logpdf terms: 1e-12 * sum |term|."""
    import math

    rng = np.random.default_rng(77)
    for n in (1, 7, 100_003):
        x = draw(rng, n)
        terms = R.TABLE[f"{name}_logpdf"](x, *params)
        rc, got = _host_sum_acc(hostlib, name, x, params)
        assert rc == 0
        assert abs(got - math.fsum(terms)) <= 1e-12 * float(np.sum(np.abs(terms))), (name, params, n, got, math.fsum(terms))


def test_host_sum_acc_exceptional_elements(hostlib):
    """Zeros, subnormals, out-of-support points, inf and NaN leave the product and add logpdf() itself: the sum is
    -inf / +inf / NaN exactly where the plain sum of the elementwise results is."""
    import math

    rng = np.random.default_rng(78)
    base = {"gamma": ((2.5, 1.3), rng.gamma(2.5, 1.0, 1000)), "beta": ((2.0, 5.0), rng.beta(2, 5, 1000)),
            "studentt": ((5.0, 0.0, 2.0), rng.standard_t(5, 1000))}
    for nm, pr, draw in _SUM_ACC_ONLY:
        base[nm] = (pr, draw(rng, 1000))
    for name, extra in (("gamma", [0.0]), ("gamma", [-1.0]), ("gamma", [5e-324, 1e-310]), ("gamma", [np.inf]), ("gamma", [np.nan]),
                        ("beta", [0.0]), ("beta", [1.0]), ("beta", [1.5]), ("beta", [1e-320]), ("beta", [np.nan, -1.0]),
                        ("studentt", [np.inf]), ("studentt", [1e200, -1e200]), ("studentt", [np.nan]),
                        ("cauchy", [np.inf]), ("cauchy", [1e300, np.nan]), ("halfcauchy", [-1.0]), ("halfcauchy", [0.0, np.inf]),
                        ("pareto", [1.0]), ("pareto", [np.inf]), ("pareto", [-2.0, np.nan]), ("inversegamma", [0.0]), ("inversegamma", [-1.0, 1e-310]),
                        ("inversegamma", [np.inf]), ("chisquared", [0.0]), ("chisquared", [-1.0]), ("chisquared", [np.inf]),
                        ("scaledinversechisquared", [0.0, 5e-324]), ("scaledinversechisquared", [np.inf, np.nan]),
                        ("betascaled", [-1.0]), ("betascaled", [3.0]), ("betascaled", [3.5, -2.0]), ("betascaled", [np.nan])):
        params, x0 = base[name]
        x = np.concatenate([x0[:500], extra, x0[500:]])
        with np.errstate(all="ignore"):
            terms = R.TABLE[f"{name}_logpdf"](x, *params)
            exp = float(np.sum(terms))
        rc, got = _host_sum_acc(hostlib, name, x, params)
        assert rc == 0
        if math.isfinite(exp):
            fin = terms[np.isfinite(terms)]
            assert abs(got - math.fsum(terms)) <= 1e-12 * float(np.sum(np.abs(fin))), (name, extra, got, exp)
        else:
            assert (math.isnan(exp) and math.isnan(got)) or got == exp, (name, extra, got, exp)
    # a == 1 (coefficient zero): x = 0 contributes xlogy0 = 0, not 0 * -inf
    x = np.concatenate([rng.gamma(1.0, 1.0, 100), [0.0]])
    rc, got = _host_sum_acc(hostlib, "gamma", x, (1.0, 2.0))
    assert rc == 0 and abs(got - math.fsum(R.gamma_logpdf(x, 1.0, 2.0))) <= 1e-12 * 200
    # coefficients above the limit decline (the kernel then sums logpdf() itself)
    assert _host_sum_acc(hostlib, "gamma", x, (5000.0, 2.0))[0] == 1
    assert _host_sum_acc(hostlib, "beta", x, (2.0, 5000.0))[0] == 1
    assert _host_sum_acc(hostlib, "studentt", x, (2e5, 0.0, 1.0))[0] == 1


@pytest.mark.parametrize("name, params, draw", _SUM_ACC_CASES, ids=[f"{c[0]}-{c[1]}" for c in _SUM_ACC_CASES])
def test_host_grad_stats_vs_oracle(hostlib, name, params, draw):
    """Grad<>::Stats (csrc/m355_grad.h: value and parameter gradients of the summed logpdf from sufficient
    statistics) against the sums of the oracle's per-element value and partials: 1e-12 * sum |term| per slot."""
    import math

    from oracle import restatement_grad as G

    rng = np.random.default_rng(79)
    for n in (1, 9, 50_001):
        x = draw(rng, n)
        val, grads = G.logpdf_grad(name, x, *params)
        out = np.empty(2 + len(params))
        p = np.ascontiguousarray(np.asarray(list(params) + [0.0] * 4, dtype=np.float64)[:4])
        assert hostlib.hm_grad_stats(DIST[name], ctypes.c_long(x.size), _ptr(np.ascontiguousarray(x)), _ptr(p), _ptr(out)) == 0
        for got, terms in zip([out[0]] + list(out[2:]), [val] + list(grads[1:])):
            terms = np.broadcast_to(terms, x.shape)
            assert abs(got - math.fsum(terms)) <= 1e-12 * max(float(np.sum(np.abs(terms))), 1.0), (name, params, n, got, math.fsum(terms))
    # exceptional elements take the graph itself: -inf value, partials as the elementwise path gives them
    x = np.concatenate([draw(rng, 100), [-1.0] if name != "studentt" else [np.inf]])
    with np.errstate(all="ignore"):
        val, grads = G.logpdf_grad(name, x, *params)
    out = np.empty(2 + len(params))
    assert hostlib.hm_grad_stats(DIST[name], ctypes.c_long(x.size), _ptr(np.ascontiguousarray(x)), _ptr(p), _ptr(out)) == 0
    with np.errstate(all="ignore"):
        exp0 = float(np.sum(val))
    assert out[0] == exp0 or (math.isnan(out[0]) and math.isnan(exp0))


def host_moment(lib, name, stat, params):
    from distributions_amd._lib import STAT

    p = np.zeros((1, 4))
    p[0, : len(params)] = params
    out = np.empty(1)
    assert lib.hm_moment(DIST[name], STAT[stat], ctypes.c_long(1), _ptr(p), _ptr(out)) == 0
    return float(out[0])


_MOMENT_CASES = H.grids() + H.moment_extra_cases()
# stats whose device body calls erfinv / erfcinv (device libm only; the host stand-in returns NaN): GPU-tested
_HOST_SKIP = {("normal", "median"), ("lognormal", "median"), ("halfnormal", "median"), ("logitnormal", "median"),
              ("truncatednormal", "median"), ("exgaussian", "mode"), ("polyagamma", "median"), ("moyal", "median"),
              ("skewnormal", "median")}


@pytest.mark.parametrize("name, sp_dist, params, sp_params, support, kw", _MOMENT_CASES,
                         ids=[f"{g[0]}-{'-'.join(str(v) for v in g[2])}" for g in _MOMENT_CASES])
def test_host_moments_vs_oracle(hostlib, name, sp_dist, params, sp_params, support, kw):
    """Mom<D>::eval (csrc/m355_moments.h, serial context) against oracle/restatement_moments.py at 1e-12 on every
    parameter point of the reference's tests (medians through ppf at the ppf tolerances of tests/harness.py)."""
    from oracle import restatement_moments as M

    for stat in M.STATS:
        if (name, stat) in _HOST_SKIP:
            continue
        if name in ("rice", "polyagamma") and stat in ("skewness", "kurtosis", "entropy", "mode", "median"):
            continue  # 3 x 1000 Bessel / series evaluations per value in a Python-driven host loop: GPU-tested
        exp = float(M.moment(name, stat, *params))
        got = host_moment(hostlib, name, stat, params)
        assert H.moment_close(name, stat, got, exp, params), (name, stat, params, got, exp)


def test_host_lgamma1p_obs(hostlib):
    """lgamma1p_obs (csrc/m355_special.h: log Gamma(x + 1) for count observations -- table below 128, Stirling from 15)
    against SciPy's gammaln: integers, the table / Stirling seams, non-integers, poles and non-finite arguments."""
    rng = np.random.default_rng(61)
    x = np.concatenate([np.arange(0, 400, dtype=np.float64), rng.uniform(0, 300, 5000), rng.uniform(14.9, 15.1, 200),
                        np.floor(np.exp(rng.uniform(np.log(100), np.log(1e15), 5000))), np.exp(rng.uniform(np.log(15), np.log(1e300), 2000)),
                        [127.0, 128.0, 127.5, 14.999999, 15.0, 1e300, 2e300, -0.5, -1.0, -2.0, -2.5, np.inf, np.nan]])
    got = vec(hostlib, "hm_lgamma1p_obs", x)
    with np.errstate(all="ignore"):
        exp = sp.gammaln(x + 1.0)
    ok = (got == exp) | (np.isnan(got) & np.isnan(exp)) | (np.abs(got - exp) <= 1e-15 * np.maximum(1.0, np.abs(exp)))
    assert np.all(ok), (x[~ok][:5], got[~ok][:5], exp[~ok][:5])


@pytest.mark.parametrize("name,params,max_bad", [
    ("gamma", (2.5, 1.0), 0), ("gamma", (0.9, 1.0), 0), ("gamma", (0.3, 1.0), 0), ("gamma", (0.05, 2.0), 0),
    ("gamma", (0.01, 1.0), 0), ("gamma", (0.5, 1e8), 0), ("gamma", (3000.0, 0.01), 0),
    ("beta", (2.0, 3.0), 0), ("beta", (0.3, 4.0), 0), ("beta", (3.0, 0.2), 0), ("beta", (0.5, 0.5), 0),
    ("beta", (0.02, 0.05), 0), ("beta", (50.0, 0.5), 0), ("beta", (200.0, 1.5), 0),
    ("studentt", (3.0, 0.5, 2.0), 2), ("chisquared", (7.0,), 0), ("inversegamma", (3.0, 2.0), 0),
    ("chisquared", (0.6,), 0), ("chisquared", (0.05,), 0), ("inversegamma", (0.3, 2.0), 0), ("inversegamma", (0.05, 1.0), 0),
    ("scaledinversechisquared", (0.7, 1.5), 0), ("halfstudentt", (5.0, 2.0), 0), ("halfstudentt", (1.0, 0.3), 0), ("halfstudentt", (50.0, 1.0), 0),
])
def test_host_tab_build(hostlib, name, params, max_bad):
    """k_tab's table build (m355_elemwise.h) restated on the host (hm_tab_build): with the range ends, the warped
    variable for shapes below 1 and the complement-aware nodes of beta, every cell of the logcdf / cdf table validates
    (|c14| + |c15| <= 1e-14 of its scale); the range is finite and increasing.  (logsf / sf of beta with alpha < 1 keep
    poisoned cells where the reference itself is a staircase -- betainc at the rounded 1 - x -- and those elements take
    the exact body: checked on the GPU, tests/test_gpu_elemwise.py::test_tabulated_kernel.)"""
    pp = np.array(list(params) + [0.0] * (4 - len(params)), dtype=np.float64)
    for fn in ("logcdf", "cdf"):
        out = np.zeros(66)
        rc = hostlib.hm_tab_build(DIST[name], FN[fn], _ptr(pp), _ptr(out))
        assert rc == 0
        assert np.isfinite(out[0]) and np.isfinite(out[1]) and out[1] > out[0]
        assert int(np.sum(~(out[2:] <= 1e-14))) <= max_bad, (fn, np.where(~(out[2:] <= 1e-14))[0])
    out = np.zeros(66)
    assert hostlib.hm_tab_build(DIST[name], FN["logsf"], _ptr(pp), _ptr(out)) == 0
    assert int(np.sum(~(out[2:] <= 1e-14))) <= (16 if name == "beta" else max_bad)
