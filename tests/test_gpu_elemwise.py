"""GPU parity tests of the elementwise path (through the C-ABI via distributions_amd):
HIP kernels vs the CPU oracle and the committed golden vectors.  Tolerances: fp64 1e-12 in the
sense |a-b| <= 1e-12*max(1,|b|), fp32 1e-5; the reference's own bar (rtol 1e-6 vs
scipy.stats, tests/helper_scipy.py) is re-run as harness A."""

import math
import os

import numpy as np
import pytest

from oracle import restatement as R
from tests import harness as H

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "elemwise.npz")
ALL_FNS = ["logpdf", "logcdf", "logsf", "pdf", "cdf", "sf", "ppf", "isf"]
FNS = {g[0]: ALL_FNS for g in H.grids()}


def mod(name):
    import importlib

    return importlib.import_module(f"distributions_amd.{name}")


CORE = ("normal", "gamma", "beta", "studentt", "poisson", "binomial")


@pytest.mark.parametrize("name, sp_dist, params, sp_params, support, kw", H.grids(), ids=H.grid_ids())
def test_harness_a_reference_bar(gpu, name, sp_dist, params, sp_params, support, kw):
    """The reference's own acceptance test (rtol 1e-6 vs scipy.stats) on the HIP path."""
    H.run_distribution_checks(mod(name), sp_dist, params, sp_params, support, **kw)


def test_golden_vectors(gpu):
    g = np.load(GOLD)
    cases = sorted({k.split("__")[0] for k in g.files})
    assert len(cases) == len(H.grids())
    for key in cases:
        name = key.split("_", 1)[1]
        params = list(g[f"{key}__params"])
        for fn in FNS[name]:
            pts = g[f"{key}__q"] if fn in ("ppf", "isf") else g[f"{key}__x"]
            exp = g[f"{key}__{fn}"]
            got = getattr(mod(name), fn)(pts, *params)
            assert got.shape == exp.shape and got.dtype == np.float64
            ok = H.ppf_close(name, fn, params, got, exp) | H.ppf_unpinned(name, fn, pts) | np.isnan(exp) & (
                fn in ("ppf", "isf"))
            if name not in CORE:  # the six core distributions take no cancellation mask: log values compared as such
                ok |= H.family_ok(name, fn, got, exp, params)
            assert np.all(ok), (key, fn, pts[~ok][:4], got[~ok][:4], exp[~ok][:4])


CONFIG2 = {
    "normal": ([0.3, 1.7], lambda r, n: r.normal(0, 1, n)),
    "gamma": ([2.5, 1.3], lambda r, n: r.gamma(2.5, 1 / 1.3, n)),
    "beta": ([2.0, 5.0], lambda r, n: r.beta(2, 5, n)),
    "studentt": ([5.0, 0.0, 2.0], lambda r, n: 2 * r.standard_t(5, n)),
}


@pytest.mark.parametrize("name", list(CONFIG2))
def test_config2_dense_parity(gpu, name):
    """SURVEY.md 8(d) config 2 inputs (seeded), 200k points, every function, fp64 1e-12."""
    params, draw = CONFIG2[name]
    rng = np.random.default_rng({"normal": 1, "gamma": 2, "beta": 3, "studentt": 4}[name])
    n = 200_003  # not a multiple of the 2048-element tile: exercises the tail path
    x = draw(rng, n)
    q = np.clip(np.random.default_rng(5).uniform(0, 1, n), 1e-12, 1 - 1e-12)
    for fn in FNS[name]:
        pts = q if fn in ("ppf", "isf") else x
        if fn in ("ppf", "isf"):
            pts = pts[:50_001]
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"{name}_{fn}"](pts, *params)
        got = getattr(mod(name), fn)(pts, *params)
        ok = H.ppf_close(name, fn, params, got, exp)
        assert np.all(ok), (name, fn, pts[~ok][:4], got[~ok][:4], exp[~ok][:4])


def test_broadcast_layouts(gpu):
    """Periodic (LDS-staged), outer, middle, full-array and scalar-x layouts give the oracle's
    broadcast result (SURVEY.md 8(d) 'broadcast params')."""
    from distributions_amd import gamma, normal, studentt

    rng = np.random.default_rng(10)
    x = rng.gamma(2.5, 1.0, (96, 1024))
    a_row = rng.uniform(0.5, 60.0, 1024)
    b_row = rng.uniform(0.2, 3.0, 1024)
    for fn in ("logpdf", "logcdf"):
        got = getattr(gamma, fn)(x, a_row, b_row)
        assert got.shape == x.shape
        assert np.all(H.close(got, R.TABLE[f"gamma_{fn}"](x, a_row, b_row)))
    # period that is not a power of two and smaller than one tile, one scalar parameter
    x2 = rng.normal(0, 2, (700, 37))
    mu_row = rng.normal(0, 1, 37)
    assert np.all(H.close(normal.logpdf(x2, mu_row, 1.3), R.normal_logpdf(x2, mu_row, 1.3)))
    # outer (R,1), middle (1,C,1) and full-size parameter arrays
    a_col = rng.uniform(0.5, 9.0, (96, 1))
    assert np.all(H.close(gamma.logpdf(x, a_col, 0.7), R.gamma_logpdf(x, a_col, 0.7)))
    x3 = rng.normal(0, 1, (5, 7, 33))
    sig_mid = rng.uniform(0.5, 2.0, (1, 7, 1))
    assert np.all(H.close(normal.logpdf(x3, 0.1, sig_mid), R.normal_logpdf(x3, 0.1, sig_mid)))
    nu_full = rng.uniform(1.0, 30.0, x3.shape)
    assert np.all(H.close(studentt.logpdf(x3, nu_full, 0.0, 2.0), R.studentt_logpdf(x3, nu_full, 0.0, 2.0)))
    # both broadcast against each other ("expand"), scalar x, 0-d result
    xa = rng.normal(0, 1, (4, 1, 8))
    mb = rng.normal(0, 1, (3, 1))
    got = normal.logpdf(xa, mb, 2.0)
    assert got.shape == (4, 3, 8)
    assert np.all(H.close(got, R.normal_logpdf(xa, mb, 2.0)))
    got = normal.logpdf(0.25, rng.normal(0, 1, 50), 1.5)
    assert got.shape == (50,)
    r0 = normal.logpdf(0.25, 0.0, 1.0)
    assert r0.shape == () and np.all(H.close(r0, R.normal_logpdf(0.25, 0.0, 1.0)))
    assert normal.logpdf(np.zeros((0, 3)), 0.0, 1.0).shape == (0, 3)


def test_dtypes_and_alignment(gpu):
    import torch

    from distributions_amd import binomial, gamma, normal, poisson

    rng = np.random.default_rng(11)
    # float32 end to end: tolerance 1e-5 (BASELINE.json)
    x32 = rng.gamma(2.5, 1.0, 10_001).astype(np.float32)
    got = gamma.logpdf(x32, np.float32(2.5), np.float32(1.3))
    assert got.dtype == np.float32
    assert np.all(H.close(got, R.gamma_logpdf(x32.astype(np.float64), 2.5, np.float64(np.float32(1.3))), 1e-5))
    # int64 observations: the support comparisons must be exact
    k = rng.integers(-3, 40, 5000)
    assert np.all(H.close(poisson.logpdf(k, 7.5), R.poisson_logpdf(k, 7.5)))
    got = binomial.logpdf(k, 30, 0.3)
    exp = R.binomial_logpdf(k, 30, 0.3)
    assert np.array_equal(np.isneginf(got), np.isneginf(exp))
    assert np.all(H.close(got, exp))
    # torch tensors in -> torch tensor out, on a misaligned view (generic kernel)
    xt = torch.from_numpy(rng.normal(0, 1, 4099)).cuda()
    view = xt[1:]
    got = normal.logpdf(view, 0.3, 1.7)
    assert isinstance(got, torch.Tensor) and got.is_cuda
    assert np.all(H.close(got.cpu().numpy(), R.normal_logpdf(view.cpu().numpy(), 0.3, 1.7)))


def test_sumlogp(gpu):
    from distributions_amd import beta, gamma, normal

    rng = np.random.default_rng(12)
    n = 3_000_001
    for m, name, params, x in (
        (normal, "normal", [0.3, 1.7], rng.normal(0, 1, n)),
        (gamma, "gamma", [2.5, 1.3], rng.gamma(2.5, 1 / 1.3, n)),
        (beta, "beta", [2.0, 5.0], rng.beta(2, 5, n)),
    ):
        terms = R.TABLE[f"{name}_logpdf"](x, *params)
        exact = math.fsum(terms)
        got = m.sum_logpdf(x, *params)
        # tolerance 1e-12 * sum |term_i|
        assert abs(got - exact) <= 1e-12 * float(np.sum(np.abs(terms))), (name, got, exact)
        assert m.sum_logpdf(x, *params) == got  # deterministic
    xr = rng.gamma(2.5, 1.0, (300, 1024))
    a_row = rng.uniform(0.5, 9.0, 1024)
    terms = R.gamma_logpdf(xr, a_row, 1.1)
    assert abs(gamma.sum_logpdf(xr, a_row, 1.1) - math.fsum(terms.ravel())) <= 1e-12 * float(np.abs(terms).sum())


def test_float32_kernels(gpu):
    """float32 in / out logpdf (compile-time dtype kernel k_scalar_f32) and float32 data into the sum-logp kernel:
    tolerance 1e-5 elementwise (BASELINE.json), 1e-12 * sum |term| for the float64 sum of the float32 data."""
    from distributions_amd import beta, gamma, normal

    rng = np.random.default_rng(44)
    n = 1_234_567  # not a multiple of the tile: vector body + scalar tail
    for m, name, params, x in ((normal, "normal", (0.3, 1.7), rng.normal(0, 1, n)), (gamma, "gamma", (2.5, 1.3), rng.gamma(2.5, 1 / 1.3, n)),
                               (beta, "beta", (2.0, 5.0), rng.beta(2, 5, n))):
        x32 = x.astype(np.float32)
        x32[::100_003] = np.float32(-1.0) if name != "normal" else np.float32(np.inf)
        exp = R.TABLE[f"{name}_logpdf"](x32.astype(np.float64), *params)
        got = m.logpdf(x32, *params)  # Python scalars are weak: float32 data gives float32 out
        assert got.dtype == np.float32
        assert np.all(H.close(got.astype(np.float64), exp, 1e-5)), name
        ok = x32 > 0 if name != "normal" else np.isfinite(x32)
        terms = exp[ok]
        tot = m.sum_logpdf(x32[ok], *params)
        assert abs(tot - math.fsum(terms)) <= 1e-12 * float(np.abs(terms).sum()), name


def test_sumlogp_product_form_edges(gpu):
    """The scalar-parameter sum-logp kernel carries sums of logarithms as products (Dist<>::SumAcc): zeros,
    subnormals, out-of-support points, inf and NaN must give what summing the elementwise results gives, and
    parameters above the coefficient limit (plain summation) the same value within the tolerance."""
    from distributions_amd import beta, gamma, studentt

    rng = np.random.default_rng(31)
    n = 1_500_007
    base = {"gamma": (gamma, (2.5, 1.3), rng.gamma(2.5, 1.0, n)), "beta": (beta, (2.0, 5.0), rng.beta(2, 5, n)),
            "studentt": (studentt, (5.0, 0.0, 2.0), 2 * rng.standard_t(5, n))}
    for name, extra in (("gamma", []), ("gamma", [0.0]), ("gamma", [-1.0]), ("gamma", [5e-324, 1e-310]), ("gamma", [np.inf]),
                        ("gamma", [np.nan]), ("beta", []), ("beta", [0.0]), ("beta", [1.0]), ("beta", [1.5]), ("beta", [1e-320]),
                        ("beta", [np.nan, -1.0]), ("studentt", []), ("studentt", [np.inf]), ("studentt", [1e200, -1e200]),
                        ("studentt", [np.nan])):
        m, params, x0 = base[name]
        x = x0.copy()
        for j, e in enumerate(extra):
            x[(j + 1) * 333_331] = e
        with np.errstate(all="ignore"):
            terms = R.TABLE[f"{name}_logpdf"](x, *params)
            exp = float(np.sum(terms))
        got = m.sum_logpdf(x, *params)
        if math.isfinite(exp):
            assert abs(got - math.fsum(terms)) <= 1e-12 * float(np.sum(np.abs(terms))), (name, extra, got, exp)
        else:
            assert (math.isnan(exp) and math.isnan(got)) or got == exp, (name, extra, got, exp)
    # the other families whose sum-logp kernel has the product form
    import importlib
    more = [("cauchy", (0.5, 2.0), 0.5 + 2.0 * rng.standard_cauchy(n), [np.inf]), ("halfcauchy", (3.5,), np.abs(3.5 * rng.standard_cauchy(n)), [-1.0]),
            ("pareto", (2.5, 1.5), 1.5 * (1 + rng.pareto(2.5, n)), [1.0]), ("inversegamma", (3.0, 2.0), 2.0 / rng.gamma(3.0, 1.0, n), [0.0]),
            ("chisquared", (7.0,), rng.chisquare(7.0, n), [-1.0]), ("scaledinversechisquared", (5.0, 0.7), 3.5 / rng.chisquare(5.0, n), [np.nan]),
            ("betascaled", (2.0, 5.0, -1.0, 3.0), -1.0 + 4.0 * rng.beta(2, 5, n), [3.5])]
    for name, params, x0, extra in more:
        m = importlib.import_module(f"distributions_amd.{name}")
        for ex in ([], extra):
            x = x0.copy()
            for j, e in enumerate(ex):
                x[(j + 1) * 333_331] = e
            with np.errstate(all="ignore"):
                terms = R.TABLE[f"{name}_logpdf"](x, *params)
                exp = float(np.sum(terms))
            got = m.sum_logpdf(x, *params)
            if math.isfinite(exp):
                assert abs(got - math.fsum(terms)) <= 1e-12 * float(np.sum(np.abs(terms))), (name, ex, got, exp)
            else:
                assert (math.isnan(exp) and math.isnan(got)) or got == exp, (name, ex, got, exp)
    # gamma(alpha = 1): x = 0 contributes 0, not 0 * -inf (quirk Q2)
    x = np.concatenate([rng.gamma(1.0, 1.0, 100_000), [0.0]])
    t = R.gamma_logpdf(x, 1.0, 2.0)
    assert abs(gamma.sum_logpdf(x, 1.0, 2.0) - math.fsum(t)) <= 1e-12 * float(np.abs(t).sum())
    # above the coefficient limit the kernel sums logpdf() itself
    for m, name, params, x in ((gamma, "gamma", (5000.0, 2.0), rng.gamma(5000.0, 0.5, 200_001)),
                               (beta, "beta", (3000.0, 2.0), rng.beta(3000, 2, 200_001)),
                               (studentt, "studentt", (4000.0, 1.0, 2.0), 1 + 2 * rng.standard_t(4000, 200_001))):
        t = R.TABLE[f"{name}_logpdf"](x, *params)
        assert abs(m.sum_logpdf(x, *params) - math.fsum(t)) <= 1e-12 * float(np.abs(t).sum()), name
    # just under the limit, data where log x is tiny (x ~ 1 - 1e-3)
    x = rng.beta(1000, 1, 1_000_003)
    t = R.beta_logpdf(x, 1000.0, 1.0)
    assert abs(beta.sum_logpdf(x, 1000.0, 1.0) - math.fsum(t)) <= 1e-12 * float(np.abs(t).sum())


def test_full_size_1e8_sampled(gpu):
    """BASELINE.json's full size: N = 1e8 fp64, every 97th element checked against the oracle."""
    import torch

    from distributions_amd import beta, gamma, normal

    n = 100_000_000
    g = torch.Generator(device="cuda").manual_seed(1)
    u = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    cases = (
        (normal, "normal", [0.3, 1.7], lambda: (u - 0.5) * 8.0),
        (gamma, "gamma", [2.5, 1.3], lambda: -torch.log1p(-u) * 2.0),
        (beta, "beta", [2.0, 5.0], lambda: u),
    )
    for m, name, params, make in cases:
        x = make()
        out = m.logpdf(x, *params)
        xs, os_ = x[::97].cpu().numpy(), out[::97].cpu().numpy()
        assert np.all(H.close(os_, R.TABLE[f"{name}_logpdf"](xs, *params))), name
        # last elements (tail handling at full size)
        xe, oe = x[-5000:].cpu().numpy(), out[-5000:].cpu().numpy()
        assert np.all(H.close(oe, R.TABLE[f"{name}_logpdf"](xe, *params))), name
        tot = m.sum_logpdf(x, *params)
        ref = out.sum(dtype=torch.float64)
        assert abs(float(tot) - float(ref)) <= 1e-12 * float(out.abs().sum()), name
        del x, out


_FAMILY_CASES = H.family_grids() + H.solver_grids()


@pytest.mark.parametrize("name, sp_dist, params, sp_params, support, kw", _FAMILY_CASES,
                         ids=[f"{g[0]}-{'-'.join(str(v) for v in g[2])}" for g in _FAMILY_CASES])
def test_families_dense_parity(gpu, name, sp_dist, params, sp_params, support, kw):
    """Closed-form / re-parameterised families: every function on 20k seeded
    points (+ the support ends and out-of-support points) against the oracle at 1e-12."""
    rng = np.random.default_rng(17)
    d = sp_dist(**sp_params)
    nx, nq = H.SLOW_ORACLE.get(name, (20_000, 20_000))
    xs = np.concatenate([np.asarray(d.rvs(nx, random_state=rng), dtype=np.float64), H.extended_vals(d, support)])
    qs = np.concatenate([rng.uniform(0, 1, nq), H.ISF_POINTS, [1e-12, 1 - 1e-12]])
    for fn in ALL_FNS:
        pts = qs if fn in ("ppf", "isf") else xs
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"{name}_{fn}"](pts, *params)
        got = getattr(mod(name), fn)(pts, *params)
        ok = (H.ppf_close(name, fn, params, got, exp) | H.family_ok(name, fn, got, exp, params) |
              H.ppf_unpinned(name, fn, pts))
        assert np.all(ok), (name, fn, params, pts[~ok][:4], got[~ok][:4], exp[~ok][:4])


@pytest.mark.parametrize("mu, sigma", [(0.0, 1.0), (0.0, 0.001), (-1.0, 1.0), (0.0, 2.0), (2.0, 0.5)])
def test_logitnormal(gpu, mu, sigma):
    """logitnormal has no SciPy counterpart (the reference tests it empirically,
    tests/test_logitnormal.py:11-40): parity against the oracle, plus the identities of the
    reference's empirical harness (cdf(ppf(q)) = q, sf + cdf = 1, isf(q) = ppf(1-q))."""
    from distributions_amd import logitnormal

    rng = np.random.default_rng(18)
    z = rng.normal(mu, sigma, 20_000)
    xs = np.concatenate([1.0 / (1.0 + np.exp(-z)), [0.0, 1.0, -1.0, 2.0, 0.5]])
    qs = np.concatenate([rng.uniform(0, 1, 20_000), H.ISF_POINTS])
    for fn in ALL_FNS:
        pts = qs if fn in ("ppf", "isf") else xs
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"logitnormal_{fn}"](pts, mu, sigma)
        got = getattr(logitnormal, fn)(pts, mu, sigma)
        ok = H.close(got, exp) | H.family_ok("logitnormal", fn, got, exp)
        assert np.all(ok), (fn, pts[~ok][:4], got[~ok][:4], exp[~ok][:4])
    q = rng.uniform(0.01, 0.99, 5000)
    assert np.allclose(logitnormal.cdf(logitnormal.ppf(q, mu, sigma), mu, sigma), q, rtol=0, atol=1e-9)
    assert np.allclose(logitnormal.sf(xs, mu, sigma) + logitnormal.cdf(xs, mu, sigma), 1.0, atol=1e-12)
    assert np.array_equal(logitnormal.isf(q, mu, sigma), logitnormal.ppf(1 - q, mu, sigma))


def _discrete_cases():
    out = [(g[0], list(g[2]), g[4]) for g in H.discrete_grids()]
    out += [(n, p, (0, 60)) for n, p in H.empirical_discrete_cases()]
    return out


@pytest.mark.parametrize("name, params, support", _discrete_cases(),
                         ids=[f"{n}-{'-'.join(map(str, p))}" for n, p, _ in _discrete_cases()])
def test_discrete_families_dense_parity(gpu, name, params, support):
    """Discrete families, every function on integer and non-integer points
    (float64 and int64 observations): values at 1e-12, ppf / isf -- the integer results of
    find_ppf_discrete (optimization.py:102-129), cycles of the iteration included -- bit-exact."""
    rng = np.random.default_rng(19)
    hi = 60 if not np.isfinite(support[1]) else int(support[1]) + 4
    xs, qs = H.discrete_points(rng, hi=max(hi, 8), n=20_000)
    m = mod(name)
    for fn in ALL_FNS:
        pts = qs if fn in ("ppf", "isf") else xs
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"{name}_{fn}"](pts, *params)
        got = getattr(m, fn)(pts, *params)
        if fn in ("ppf", "isf"):
            ok = (got == exp) | (np.isnan(got) & np.isnan(exp))
        else:
            ok = H.close(got, exp) | H.family_ok(name, fn, got, exp, params)
        assert np.all(ok), (name, fn, params, pts[~ok][:4], got[~ok][:4], exp[~ok][:4])
    k = rng.integers(-3, max(hi, 8), 5000)
    with np.errstate(all="ignore"):
        exp = R.TABLE[f"{name}_logpdf"](k.astype(np.float64), *params)
    got = m.logpdf(k, *params)
    assert np.array_equal(np.isneginf(got), np.isneginf(exp)) and np.all(H.close(got, exp))
    fin = np.isfinite(exp)
    if fin.any():
        tot = m.sum_logpdf(k[fin], *params)
        assert abs(tot - math.fsum(exp[fin])) <= 1e-12 * float(np.abs(exp[fin]).sum())


@pytest.mark.parametrize("n, p", [(1000.0, 0.3), (300.0, 0.5), (1e5, 0.02), (1e7, 0.7), (1e12, 0.5)])
def test_binomial_logpmf_large_counts(gpu, n, p):
    """binomial.logpdf where every lane of a wavefront has x >= 128 and n - x >= 128 takes the paired log-factorial
    (lgamma_obs_pair_big: table logarithms, one shared reciprocal, binomial.py:66-75), wavefronts with a small count, a
    value outside the support or a non-integer among their lanes the general body; both against the oracle, whose own
    three gammaln terms carry ~eps * gammaln(n + 1) of rounding -- hence the bar 1e-12 max(1, |b|) + 4 eps gammaln(n + 1)."""
    from scipy.special import gammaln

    from distributions_amd import binomial

    rng = np.random.default_rng(int(n) % 1000 + 3)
    x = rng.binomial(int(n), p, 300_000).astype(np.float64)
    mixed = x.copy()
    mixed[::37] = rng.integers(0, 130, mixed[::37].size)  # small counts: table branch in some wavefronts
    mixed[5::53] = -1.0
    mixed[7::59] = n + 1.0
    mixed[11::61] += 0.5
    for pts in (x, mixed, x.astype(np.int64)):
        with np.errstate(all="ignore"):
            exp = R.TABLE["binomial_logpdf"](np.asarray(pts, dtype=np.float64), n, p)
        got = binomial.logpdf(pts, n, p)
        assert np.array_equal(np.isneginf(got), np.isneginf(exp)) and np.array_equal(np.isnan(got), np.isnan(exp))
        fin = np.isfinite(exp)
        err = np.abs(got[fin] - exp[fin])
        assert np.all(err <= 1e-12 * np.maximum(1.0, np.abs(exp[fin])) + 4 * 2.3e-16 * gammaln(n + 1.0)), (n, p, err.max())
    if n <= 1e7:  # against exact integer arithmetic: log of the exact rational pmf
        import mpmath as mp

        mp.mp.dps = 40
        for xi in x[:40]:
            ref = mp.log(mp.binomial(int(n), int(xi))) + xi * mp.log(p) + (n - xi) * mp.log1p(-mp.mpf(p))
            gi = binomial.logpdf(np.array([xi] * 64), n, p)[0]
            assert abs(gi - ref) <= 1e-12 * max(1.0, abs(ref)) + 4 * 2.3e-16 * float(gammaln(n + 1.0)), (n, p, xi)


@pytest.mark.parametrize("name, params", [("gamma", (2.5, 1.3)), ("gamma", (1.0, 0.2)), ("gamma", (400.0, 2.0)),
                                          ("gamma", (8e3, 1e-3)), ("beta", (2.0, 5.0)), ("beta", (1.0, 1.0)),
                                          ("beta", (60.0, 1.5)), ("beta", (150.0, 300.0)), ("gamma", (0.3, 1.0)),
                                          ("beta", (0.5, 0.5)), ("gamma", (3e5, 1e-3)), ("beta", (300.0, 800.0)),
                                          ("studentt", (5.0, 0.0, 2.0)), ("studentt", (1.0, 1.0, 0.5)),
                                          ("studentt", (30.0, -3.0, 1e-3)), ("studentt", (300.0, 0.0, 1.0)),
                                          ("studentt", (0.5, 0.0, 1.0)), ("gamma", (0.05, 2.0)), ("beta", (0.3, 4.0)),
                                          ("beta", (3.0, 0.2)), ("gamma", (0.005, 1.0)), ("halfstudentt", (5.0, 2.0)),
                                          ("halfstudentt", (1.0, 0.5)), ("halfstudentt", (40.0, 1e-2)),
                                          ("halfstudentt", (80.0, 1.0))])
def test_tabulated_kernel(gpu, name, params):
    """k_tab (per-launch Chebyshev tables of logcdf / logsf / cdf / sf for scalar parameters, m355_elemwise.h), forced
    onto a small input through m355_set_tab_min_n: against the oracle at the parity bar and against the exact kernels at
    5e-13 max(1, |b|) (the exact bodies' own noise at large parameters is what the table smooths); the special values (NaN, infinities, the support's ends, points outside it, the far tails beyond
    the table) must come out as the exact kernels give them, bit for bit; shapes below 0.01 and very large ones are not
    tabulated at all (Dist<>::tab_ok); shapes in [0.01, 1) go through the asinh warp of the table variable, for the cdf
    family only (Dist<>::tab_ok_fn)."""
    from distributions_amd import _lib

    L = _lib.lib()
    nan, inf = np.nan, np.inf
    rng = np.random.default_rng(33)
    m = mod(name)
    n = 40_000 + 37  # not a multiple of the tile
    if name == "gamma":
        x = rng.gamma(params[0], 1.0 / params[1], n)
        special = np.array([nan, inf, -inf, 0.0, -1.0, 5e-324, 1e-300, 1e300, x.min() * 1e-6, x.max() * 30])
    elif name == "studentt":
        nu, mu, sigma = params
        x = mu + sigma * rng.standard_t(nu, n)
        x[100:3000] = mu + sigma * rng.normal(0, 1e-3, 2900)  # the band around the centre and its edges
        special = np.array([nan, inf, -inf, mu, mu + 1e-9 * sigma, mu - 1e-7 * sigma, mu + 1e30 * sigma, mu - 1e30 * sigma])
    elif name == "halfstudentt":  # tabulated in the reference's own rounded nu / (xn^2 + nu) (Dist<M355_HALFSTUDENTT>)
        nu, sigma = params
        x = sigma * np.abs(rng.standard_t(nu, n))
        x[100:3000] = sigma * np.abs(rng.normal(0, 2e-3, 2900))  # the band next to 0 and its edge
        special = np.array([nan, inf, -inf, 0.0, -1.0, 5e-324, 1e-9 * sigma, 1e30 * sigma])
    else:
        x = rng.beta(params[0], params[1], n)
        special = np.array([nan, inf, -inf, 0.0, 1.0, -0.5, 1.5, 5e-324, 1e-300, 1 - 2.0**-20, x.min() * 1e-6])
    x[: special.size] = special
    if name == "gamma":
        tabulated = 0.01 <= params[0] <= 1e4
    elif name in ("studentt", "halfstudentt"):
        tabulated = 1.0 <= params[0] <= 50.0
    else:
        tabulated = min(params) >= 0.01 and sum(params) <= 500.0
    # quantiles: gamma from alpha = 0.05, beta from a = 0.1 with b >= 1 (shapes below 1 as tables of log(ppf): tab_ppf_log)
    tabulated_q = tabulated and (name in ("studentt", "halfstudentt") or (name == "gamma" and params[0] >= 0.05) or
                                 (name == "beta" and params[0] >= 0.1 and params[1] >= 1.0))
    tabulated_all = tabulated
    for fn in ("logcdf", "logsf", "cdf", "sf"):
        # (beta.logsf / sf with alpha < 1/2 are left to the exact kernels: Dist<M355_BETA>::tab_ok_fn)
        tabulated = tabulated_all and not (name == "beta" and fn in ("logsf", "sf") and params[0] < 0.5)
        old = L.m355_set_tab_min_n(2**62)
        try:
            exact = getattr(m, fn)(x, *params)
            L.m355_set_tab_min_n(1)
            tab = getattr(m, fn)(x, *params)
        finally:
            L.m355_set_tab_min_n(old)
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"{name}_{fn}"](x, *params)
        assert np.array_equal(np.isnan(tab), np.isnan(exact))
        sp_ = slice(0, 7)  # NaN, infinities, the ends of the support, points outside it
        assert np.array_equal(tab[sp_], exact[sp_], equal_nan=True), (fn, x[sp_], tab[sp_], exact[sp_])
        if tabulated:  # (the exact kernels' own parity is the business of the tests above)
            ok = H.close(tab, exp) | H.family_ok(name, fn, tab, exp, params)
            # misses carry the exact body's value, whose parity is the business of the tests above; two regions where
            # SciPy's betainc (the oracle) is itself off are misses by construction: subnormal x, and logsf / sf of
            # beta below x ~ 1e-6, where the reference takes betainc at the rounded 1 - x and SciPy is inaccurate
            # next to 1 (beta(1/2, 1/2), 1 - x = 1 - 2^-53: sf = 1 - 6.7e-9 exactly, SciPy 1 - 9.5e-9)
            ok |= tab == exact
            assert np.all(ok), (name, fn, params, x[~ok][:4], tab[~ok][:4], exp[~ok][:4])
        fin = np.isfinite(exact)
        assert np.all(np.abs(tab[fin] - exact[fin]) <= 5e-13 * np.maximum(1.0, np.abs(exact[fin])))
        if tabulated:
            assert np.mean(tab == exact) < 0.5  # the table really was used
        else:
            assert np.array_equal(tab, exact, equal_nan=True)
    # quantiles: tabulated in logit(q) over [1e-9, 1 - 1e-9]; beyond, q = 0, 1, NaN and q outside [0, 1] from the exact body
    q = rng.uniform(0, 1, n)
    q[:2000] = np.exp(rng.uniform(-20, 0, 2000))        # lower tail inside the table
    q[2000:4000] = 1 - np.exp(rng.uniform(-20, 0, 2000))  # upper tail inside the table (1 - q exact)
    specq = np.array([nan, 0.0, 1.0, -0.1, 1.1, 1e-300, 1e-12, 1 - 2.0**-53, 1 - 1e-12, 0.5, 0.5 + 1e-15])
    q[4000 : 4000 + specq.size] = specq
    for fn in ("ppf", "isf"):
        old = L.m355_set_tab_min_n(2**62)
        try:
            exact = getattr(m, fn)(q, *params)
            L.m355_set_tab_min_n(1)
            tab = getattr(m, fn)(q, *params)
        finally:
            L.m355_set_tab_min_n(old)
        sp_ = slice(4000, 4000 + (specq.size if name == "studentt" else 9))  # (the median is a miss for studentt only)
        assert np.array_equal(tab[sp_], exact[sp_], equal_nan=True), (fn, specq, tab[sp_], exact[sp_])
        assert np.array_equal(np.isnan(tab), np.isnan(exact))
        fin = np.isfinite(exact)
        if name in ("studentt", "halfstudentt"):  # the exact body follows the reference's quantised sqrt(1/b - 1)
            # (the window is modelled at the second argument: next to q = 0 halfstudentt's reference value can be 0
            # itself -- b rounds to 1 -- so it is taken at whichever of the two is the smooth one)
            assert np.all(H.ppf_close(name, fn, params, tab[fin], exact[fin]) | H.ppf_close(name, fn, params, exact[fin], tab[fin])), fn
        else:  # (a table of log(ppf) carries its absolute error, ~1e-14 max(1, |log x|), as a relative error of x)
            with np.errstate(all="ignore"):
                lg = np.where(exact[fin] > 0, np.abs(np.log(np.where(exact[fin] > 0, exact[fin], 1.0))), 0.0)
            assert np.all(np.abs(tab[fin] - exact[fin]) <= (2e-13 + 3e-14 * lg) * np.abs(exact[fin])), fn
        if tabulated_q:
            with np.errstate(all="ignore"):
                exp = R.TABLE[f"{name}_{fn}"](q, *params)
            ok = H.ppf_close(name, fn, params, tab, exp) | H.ppf_close(name, fn, params, exp, tab) | H.ppf_unpinned(name, fn, q)
            assert np.all(ok | ~np.isfinite(exp)), (name, fn, params, q[~ok][:4], tab[~ok][:4], exp[~ok][:4])
            assert np.mean(tab == exact) < 0.5
        else:
            assert np.array_equal(tab, exact, equal_nan=True)
    assert L.m355_set_tab_min_n(-1) == old and L.m355_set_tab_min_n(old) == 2**21


@pytest.mark.parametrize("name, params", [("chisquared", (7.0,)), ("chisquared", (300.0,)), ("inversegamma", (3.0, 2.0)),
                                          ("inversegamma", (40.0, 1e3)), ("scaledinversechisquared", (5.0, 0.7)),
                                          ("betascaled", (2.0, 5.0, -1.0, 3.0)), ("betascaled", (30.0, 1.5, 10.0, 10.5)),
                                          ("chisquared", (0.6,)), ("inversegamma", (0.3, 2.0)),
                                          ("scaledinversechisquared", (0.7, 1.5))])
def test_tabulated_kernel_relatives(gpu, name, params):
    """k_tab for the families of m355_dists2.h that share gamma's and beta's table variables (chisquared, inversegamma,
    scaledinversechisquared: log of the scaled argument; betascaled: logit of the standardised one): all six functions
    against the exact kernels (5e-13 max(1, |b|), quantiles 2e-13 relative) and the oracle (parity bar), NaN / infinities /
    out-of-support points bit for bit, and the table really in use."""
    from distributions_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(35)
    m = mod(name)
    n = 30_000 + 11
    if name == "chisquared":
        x = rng.chisquare(params[0], n)
    elif name == "inversegamma":
        x = params[1] / rng.gamma(params[0], 1.0, n)
    elif name == "scaledinversechisquared":
        x = params[0] * params[1] / rng.chisquare(params[0], n)
    else:
        x = params[2] + (params[3] - params[2]) * rng.beta(params[0], params[1], n)
    lo, hi = (params[2], params[3]) if name == "betascaled" else (0.0, np.inf)
    special = np.array([np.nan, np.inf, -np.inf, lo, lo - 1.0, hi if np.isfinite(hi) else 1e300, (hi if np.isfinite(hi) else 1e300) + 1.0])
    x[: special.size] = special
    q = rng.uniform(0, 1, n)
    q[:1000] = np.exp(rng.uniform(-18, 0, 1000))
    q[1000:2000] = 1 - np.exp(rng.uniform(-18, 0, 1000))
    specq = np.array([np.nan, 0.0, 1.0, -0.1, 1.1])
    q[2000:2005] = specq
    for fn in ("logcdf", "logsf", "cdf", "sf", "ppf", "isf"):
        quant = fn in ("ppf", "isf")
        arg = q if quant else x
        old = L.m355_set_tab_min_n(2**62)
        try:
            exact = getattr(m, fn)(arg, *params)
            L.m355_set_tab_min_n(1)
            tab = getattr(m, fn)(arg, *params)
        finally:
            L.m355_set_tab_min_n(old)
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"{name}_{fn}"](arg, *params)
        assert np.array_equal(np.isnan(tab), np.isnan(exact)), fn
        sp_ = slice(2000, 2005) if quant else slice(0, special.size)
        assert np.array_equal(tab[sp_], exact[sp_], equal_nan=True), (fn, tab[sp_], exact[sp_])
        fin = np.isfinite(exact)
        if quant:
            with np.errstate(all="ignore"):  # (tables of log(ppf) for shapes below 1: relative error ~1e-14 max(1, |log x|))
                lg = np.where(exact[fin] > 0, np.abs(np.log(np.where(exact[fin] > 0, exact[fin], 1.0))), 0.0)
            assert np.all(np.abs(tab[fin] - exact[fin]) <= (2e-13 + 3e-14 * lg) * np.maximum(np.abs(exact[fin]), 1e-300)
                          + 1e-13 * (hi - lo if np.isfinite(hi) else 0.0)), fn
            ok = H.ppf_close(name, fn, params, tab, exp) | H.ppf_unpinned(name, fn, arg) | ~np.isfinite(exp) | (tab == exact)
        else:
            assert np.all(np.abs(tab[fin] - exact[fin]) <= 5e-13 * np.maximum(1.0, np.abs(exact[fin]))), fn
            ok = H.close(tab, exp) | H.family_ok(name, fn, tab, exp, params) | (tab == exact)
        assert np.all(ok), (name, fn, params, arg[~ok][:4], tab[~ok][:4], exp[~ok][:4])
        # (betascaled on [10, 10.5]: x carries ulp(10) / 0.5 = 4e-15 of the standardised variable, times the slope ~ a = 30:
        # half the cells cannot show 1e-14 and are poisoned -- those elements are misses, by design)
        assert np.mean(tab == exact) < (0.9 if params[-1] == 10.5 else 0.5), fn


@pytest.mark.parametrize("name, params", [("poisson", (100.0,)), ("poisson", (0.7,)), ("binomial", (1000.0, 0.3)),
                                          ("binomial", (12.0, 0.9)), ("negativebinomial", (10.0, 0.1)),
                                          ("betabinomial", (100.0, 20.0, 20.0)), ("hypergeometric", (50.0, 20.0, 10.0)),
                                          ("zi_poisson", (0.7, 10.0)), ("zi_binomial", (0.4, 60.0, 0.2)),
                                          ("zi_negativebinomial", (0.8, 5.0, 0.3)), ("poisson", (1e7,))])
def test_discrete_lookup_kernel(gpu, name, params):
    """k_dtab / k_dtab_ppf (cdf / sf / logcdf / logsf of a discrete family answered from a per-launch table of the exact
    body over a window of integers, ppf / isf by the reference's search on the tabulated values, m355_elemwise.h), forced
    onto a small input: bit-identical to the exact kernels for counts
    inside and outside the window, non-integers, negatives, NaN and infinities, float64 and int64 observations; a
    window that does not fit (poisson(1e7): 14 sd > 2048 values) leaves the exact kernels in charge."""
    from distributions_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(41)
    m = mod(name)
    n = 30_000 + 11
    k = np.asarray(m.rvs(*params, size=n, random_state=7) if hasattr(m, "rvs") else rng.integers(0, 60, n), dtype=np.float64)
    k[:12] = [np.nan, np.inf, -np.inf, -1.0, 0.0, 0.5, 1e18, 3.0, 2.5, k.max() * 50 + 7, -0.0, 1.0]
    for obs in (k, np.where(np.isfinite(k), k, 0).astype(np.int64)):
        for fn in ("cdf", "sf", "logcdf", "logsf"):
            old = L.m355_set_tab_min_n(2**62)
            try:
                exact = getattr(m, fn)(obs, *params)
                L.m355_set_tab_min_n(1)
                tab = getattr(m, fn)(obs, *params)
            finally:
                L.m355_set_tab_min_n(old)
            assert np.array_equal(tab, exact, equal_nan=True), (name, fn, obs[~((tab == exact) | (np.isnan(tab) & np.isnan(exact)))][:5])
    # quantiles: find_ppf_discrete on the tabulated cdf / pmf -- the same iterates, bit for bit
    q = rng.uniform(0, 1, n)
    q[:4000] = np.exp(rng.uniform(-25, 0, 4000))
    q[4000:8000] = 1 - np.exp(rng.uniform(-25, 0, 4000))
    q[8000:8008] = [np.nan, 0.0, 1.0, -0.5, 1.5, 1e-300, 1 - 2.0**-53, 0.5]
    for fn in ("ppf", "isf"):
        old = L.m355_set_tab_min_n(2**62)
        try:
            exact = getattr(m, fn)(q, *params)
            L.m355_set_tab_min_n(1)
            tab = getattr(m, fn)(q, *params)
        finally:
            L.m355_set_tab_min_n(old)
        assert np.array_equal(tab, exact, equal_nan=True), (name, fn, q[~((tab == exact) | (np.isnan(tab) & np.isnan(exact)))][:5])


@pytest.mark.parametrize("P", [48, 40])
@pytest.mark.parametrize("name", ["poisson", "binomial", "negativebinomial", "betabinomial", "hypergeometric", "zi_poisson",
                                  "zi_binomial", "zi_negativebinomial"])
def test_discrete_lookup_per_column(gpu, name, P):
    """k_dcol_ranges + k_dcol_build + k_dcolg (P a multiple of 16: column groups, windows in LDS; quantiles by
    find_ppf_discrete on the tabulated cdf / pmf) or k_dcol (any other P: cdf family only) -- row-periodic parameters, one
    window of exact values per column in the call's workspace, m355_elemwise.h -- forced onto a small input: bit-identical
    to the exact kernels, for float64 and int64 observations, counts inside and outside the windows, non-integers, NaN,
    infinities, and columns whose window does not fit or whose parameters are out of domain (left to the exact bodies)."""
    from distributions_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(43)
    m = mod(name)
    R = 700
    pp = {
        "poisson": (np.concatenate([rng.uniform(0.2, 300.0, P - 2), [1e7, -1.0]]),),
        "binomial": (np.concatenate([rng.integers(1, 2000, P - 1), [5_000_000]]).astype(np.float64), rng.uniform(0.02, 0.98, P)),
        "negativebinomial": (rng.uniform(0.5, 40.0, P), rng.uniform(0.05, 0.9, P)),
        "betabinomial": (rng.integers(1, 300, P).astype(np.float64), rng.uniform(0.5, 30.0, P), rng.uniform(0.5, 30.0, P)),
        "hypergeometric": (np.full(P, 500.0), rng.integers(1, 400, P).astype(np.float64), rng.integers(1, 300, P).astype(np.float64)),
        "zi_poisson": (rng.uniform(0.1, 0.9, P), rng.uniform(0.5, 200.0, P)),
        "zi_binomial": (rng.uniform(0.1, 0.9, P), rng.integers(1, 500, P).astype(np.float64), rng.uniform(0.05, 0.95, P)),
        "zi_negativebinomial": (rng.uniform(0.1, 0.9, P), rng.uniform(0.5, 30.0, P), rng.uniform(0.05, 0.9, P)),
    }[name]
    with np.errstate(all="ignore"):
        k = np.asarray(m.rvs(*[np.where(np.isfinite(a) & (a > 0), a, 1.0) if j == 0 and name == "poisson" else a for j, a in enumerate(pp)],
                             size=(R, P), random_state=9), dtype=np.float64)
    k[0, :12] = [np.nan, np.inf, -np.inf, -1.0, 0.0, 0.5, 1e18, 3.0, 2.5, 1e6, -0.0, 1.0]
    k[1] = k[2] * 3 + 50  # far in the upper tails: outside most windows
    for obs in (k, np.where(np.isfinite(k), np.clip(k, -1e15, 1e15), 0).astype(np.int64)):
        for fn in ("cdf", "sf", "logcdf", "logsf"):
            old = L.m355_set_tab_min_n(2**62)
            try:
                with np.errstate(all="ignore"):
                    exact = getattr(m, fn)(obs, *pp)
                L.m355_set_tab_min_n(1)
                with np.errstate(all="ignore"):
                    tab = getattr(m, fn)(obs, *pp)
            finally:
                L.m355_set_tab_min_n(old)
            assert tab.shape == (R, P)
            assert np.array_equal(tab, exact, equal_nan=True), (name, fn, np.argwhere(~((tab == exact) | (np.isnan(tab) & np.isnan(exact))))[:5])
    # quantiles: the reference's search on the tabulated values -- the same iterates, bit for bit
    q = rng.uniform(0, 1, (R, P))
    q[:40] = np.exp(rng.uniform(-25, 0, (40, P)))
    q[40:80] = 1 - np.exp(rng.uniform(-25, 0, (40, P)))
    q[80, :8] = [np.nan, 0.0, 1.0, -0.5, 1.5, 1e-300, 1 - 2.0**-53, 0.5]
    for fn in ("ppf", "isf"):
        old = L.m355_set_tab_min_n(2**62)
        try:
            with np.errstate(all="ignore"):
                exact = getattr(m, fn)(q, *pp)
            L.m355_set_tab_min_n(1)
            with np.errstate(all="ignore"):
                tab = getattr(m, fn)(q, *pp)
        finally:
            L.m355_set_tab_min_n(old)
        assert np.array_equal(tab, exact, equal_nan=True), (name, fn, np.argwhere(~((tab == exact) | (np.isnan(tab) & np.isnan(exact))))[:5])
    # the kernels really ran: after a call through the C-ABI with a NaN-filled workspace the column headers are in it
    import torch

    from distributions_amd import raw

    ws = torch.full((8 + P * (2 + 2 * 2048),), float("nan"), dtype=torch.float64, device=gpu)
    xd = torch.from_numpy(k).to(gpu)
    out = torch.empty_like(xd)
    old = L.m355_set_tab_min_n(1)
    try:
        raw.elemwise_into(name, "logcdf", xd, tuple(torch.from_numpy(np.ascontiguousarray(a)).to(gpu) for a in pp), out,
                          workspace=ws.view(torch.uint8))
        torch.cuda.synchronize()
    finally:
        L.m355_set_tab_min_n(old)
    hdr = ws[8 : 8 + 2 * P].cpu().numpy().reshape(P, 2)
    assert int(ws[:1].view(torch.int32)[0].item()) == int(hdr[:, 1].max())  # the tables' stride: the longest window
    assert not np.isnan(hdr).any() and (hdr[:, 1] > 0).sum() >= P // 2, hdr[:4]
    if name == "poisson":
        assert hdr[-1, 1] == -1.0 and hdr[-2, 1] == -1.0  # mu = -1 and mu = 1e7 (a window of > 2048 values): no table


@pytest.mark.parametrize("h, z", H.POLYAGAMMA_CASES)
def test_polyagamma(gpu, h, z):
    """polyagamma: no SciPy counterpart -- every function against the oracle (series density,
    500-point trapezoid cdf, the module's own Newton ppf), plus cdf(ppf(q)) = q and the bounds of the reference's
    test (tests/test_polyagamma.py:60-66)."""
    from distributions_amd import polyagamma

    xs, qs = H.polyagamma_points(np.random.default_rng(42), h, z)
    for fn in ALL_FNS:
        pts = qs if fn in ("ppf", "isf") else xs
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"polyagamma_{fn}"](pts, h, z)
        got = getattr(polyagamma, fn)(pts, h, z)
        tol = 1e-9 if fn in ("ppf", "isf") else 1e-12  # Newton on a cdf that carries ~1e-14 of summation noise
        ok = H.close(got, exp, tol) | H.family_ok("polyagamma", fn, got, exp)
        assert np.all(ok), (fn, h, z, pts[~ok][:4], got[~ok][:4], exp[~ok][:4])
    q = np.linspace(0.05, 0.95, 10)
    assert np.allclose(polyagamma.cdf(polyagamma.ppf(q, h, z), h, z), q, rtol=0, atol=1e-7)
    ext = np.array([0.0, np.inf, -1.0, -2.0, np.inf, np.inf])
    assert np.array_equal(polyagamma.cdf(ext, h, z), [0.0, 1.0, 0.0, 0.0, 1.0, 1.0])
    # rvs: KS against the module's own cdf, mean against the closed form (tests/test_polyagamma.py:34-41)
    from scipy import stats

    x = polyagamma.rvs(h, z, size=50_000, random_state=8)
    assert x.dtype == np.float64 and np.all(x > 0)
    assert stats.kstest(x[:4000], lambda v: R.TABLE["polyagamma_cdf"](v, h, z)).pvalue > 1e-3
    mean = h / 4.0 if z == 0 else h / (2 * z) * np.tanh(z / 2)
    assert abs(x.mean() - mean) <= 3e-2 * mean


@pytest.mark.parametrize("name, fn, params", [("gamma", "logcdf", (2.5, 1.3)), ("beta", "ppf", (2.0, 5.0)),
                                               ("studentt", "logsf", (5.0, 0.0, 2.0)), ("poisson", "cdf", (7.5,)),
                                               ("betabinomial", "cdf", (20.0, 0.5, 0.5)), ("wald", "ppf", (2.0, 10.0))])
def test_binned_kernel_equals_scalar_kernel(gpu, name, fn, params):
    """k_binned (LDS sort of each 2048-element tile by x, evaluation in sorted order, un-permuted stores) returns
    bit for bit what k_scalar returns: inputs of >= 256 tiles take the binned path, smaller slices the plain one."""
    rng = np.random.default_rng(51)
    n = 520 * 2048 + 777  # a ragged tail after the last full tile
    if fn in ("ppf", "isf"):
        x = rng.uniform(0, 1, n)
        x[:7] = [0.0, 1.0, -0.5, 1.5, np.nan, 0.5, 1e-300]
    elif name in ("poisson", "betabinomial"):
        x = rng.integers(-2, 30, n).astype(np.float64)
    else:
        x = np.abs(rng.standard_cauchy(n)) if name != "studentt" else rng.standard_cauchy(n) * 3
        x[:6] = [0.0, np.inf, -np.inf, np.nan, -1.0, 1e-300]
    if name == "beta" and fn not in ("ppf", "isf"):
        x = rng.uniform(-0.1, 1.1, n)
    m = mod(name)
    big = getattr(m, fn)(x, *params)
    parts = np.concatenate([getattr(m, fn)(x[i:i + 300_000], *params) for i in range(0, n, 300_000)])
    assert np.array_equal(big, parts, equal_nan=True)
    if name == "poisson":  # int64 observations through the binned path
        k = x.astype(np.int64)
        assert np.array_equal(getattr(m, fn)(k, *params), big, equal_nan=True)


@pytest.mark.parametrize("name, fn, P", [("gamma", "logcdf", 1024), ("gamma", "ppf", 1024), ("beta", "logsf", 64), ("beta", "ppf", 8),
                                         ("studentt", "logcdf", 1000), ("studentt", "isf", 40), ("gamma", "cdf", 1001)])
def test_colbinned_kernel_equals_periodic_kernel(gpu, name, fn, P):
    """k_colbinned (row-periodic parameters: 512 x 8 tiles, each column sorted by x in LDS and evaluated with its one
    parameter set) returns bit for bit what k_periodic returns: inputs of >= 4 blocks of 512 rows take the sorted path
    (with a k_periodic launch for the rows left over), slices of fewer rows the plain one; P = 1001 is not a
    multiple of 8 and stays on k_periodic altogether.  Both against the oracle at 1e-12 too."""
    rng = np.random.default_rng(52)
    rows = 2600 if P >= 1000 else (5000 if P >= 40 else 20_000)
    tail = 17 if P > 17 else 3  # a partial row after the last full one
    n = rows * P + tail
    if fn in ("ppf", "isf"):
        x = rng.uniform(0, 1, n)
        x[:7] = [0.0, 1.0, -0.5, 1.5, np.nan, 0.5, 1e-300]
    elif name == "beta":
        x = rng.uniform(-0.05, 1.05, n)
    else:
        x = np.abs(rng.standard_cauchy(n)) * 2 if name != "studentt" else rng.standard_cauchy(n) * 3
        x[:6] = [0.0, np.inf, -np.inf, np.nan, -1.0, 1e-300]
    params = {"gamma": (rng.uniform(0.3, 12.0, P), rng.uniform(0.3, 3.0, P)), "beta": (rng.uniform(0.4, 9.0, P), 2.5),
              "studentt": (rng.uniform(1.5, 40.0, P), 0.3, rng.uniform(0.5, 2.0, P))}[name]
    m = mod(name)
    # the library sees x flat with parameters of period P: call through a (rows, P) view for the full rows and
    # evaluate the partial row separately
    body = x[: rows * P].reshape(rows, P)
    big = getattr(m, fn)(body, *params)
    step = 1500  # < 4 * 512 rows: k_periodic
    parts = np.concatenate([getattr(m, fn)(body[i:i + step], *params) for i in range(0, rows, step)])
    assert np.array_equal(big, parts, equal_nan=True)
    with np.errstate(all="ignore"):
        exp = R.TABLE[f"{name}_{fn}"](body, *params)
    ok = H.ppf_close(name, fn, None, big, exp) | H.ppf_unpinned(name, fn, body)
    assert np.all(ok), (name, fn, body[~ok][:4], big[~ok][:4], exp[~ok][:4])
    # flat call with the partial row: the operand description (div = 1, mod = P) is what _dispatch builds for a
    # 1-d x of length n against np.resize-d parameters -- here through raw operands
    import ctypes

    import torch

    from distributions_amd import _lib
    from distributions_amd._lib import DIST, F64, FN, Operand

    xt = torch.from_numpy(x).to(gpu)
    out = torch.empty(n, dtype=torch.float64, device=gpu)
    keep = [torch.from_numpy(np.ascontiguousarray(p)).to(gpu) if np.ndim(p) else p for p in params]
    ops = (Operand * (1 + len(params)))()
    ops[0].ptr, ops[0].div, ops[0].mod, ops[0].dtype = xt.data_ptr(), 1, n, F64
    for j, p in enumerate(keep, start=1):
        if isinstance(p, float):
            ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = None, p, 1, 1, F64
        else:
            ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = p.data_ptr(), 0.0, 1, P, F64
    at = lambda k: ctypes.cast(ctypes.byref(ops, k * ctypes.sizeof(Operand)), ctypes.POINTER(Operand))
    wsb = torch.empty(int(_lib.lib().m355_elemwise_workspace_bytes()), dtype=torch.uint8, device=gpu)
    _lib.check(_lib.lib().m355_elemwise_ws(DIST[name], FN[fn], F64, n, at(0), len(params), at(1), out.data_ptr(),
                                           wsb.data_ptr(), wsb.numel(), torch.cuda.current_stream(gpu).cuda_stream))
    flat = out.cpu().numpy()
    assert np.array_equal(flat[: rows * P].reshape(rows, P), big, equal_nan=True)
    tail_params = [np.asarray(p)[:tail] if np.ndim(p) else p for p in params]
    assert np.array_equal(flat[rows * P:], getattr(m, fn)(x[rows * P:], *tail_params), equal_nan=True)


@pytest.mark.parametrize("name, fn, P", [("gamma", "logcdf", 1024), ("gamma", "ppf", 32), ("gamma", "sf", 48), ("beta", "logsf", 64),
                                         ("beta", "cdf", 16), ("beta", "isf", 32), ("studentt", "logcdf", 32),
                                         ("studentt", "ppf", 16), ("studentt", "logsf", 1024), ("inversegamma", "logcdf", 32),
                                         ("inversegamma", "ppf", 16), ("halfstudentt", "logcdf", 32), ("halfstudentt", "sf", 16),
                                         ("halfstudentt", "ppf", 16)])
def test_coltab_kernel(gpu, name, fn, P):
    """k_coltab (row-periodic parameters with P a multiple of 16: one Chebyshev table per column, 16 columns per
    workgroup), forced onto a small input through m355_set_tab_min_n: against the oracle at the parity bar and against
    the exact kernels (k_colbinned / k_periodic) at 5e-13 max(1, |b|) / 2e-13 relative for quantiles; special values
    bit for bit; columns whose parameters Dist<>::tab_ok declines (the first three) come out exactly as the exact
    kernels give them; the rows need not fill the last block of 256, and a partial last row goes through k_periodic."""
    import ctypes

    import torch

    from distributions_amd import _lib
    from distributions_amd._lib import DIST, F64, FN, Operand

    L = _lib.lib()
    nan, inf = np.nan, np.inf
    rng = np.random.default_rng(77)
    rows = 300 if P >= 1000 else 1111
    tail = 5
    n = rows * P + tail
    quant = fn in ("ppf", "isf")
    if name == "gamma":
        params = (rng.uniform(1.0, 30.0, P), rng.uniform(0.2, 5.0, P))
        params[0][P // 2:] = rng.uniform(0.05, 1.0, P - P // 2)  # warped tables / tables of log(ppf)
        params[0][:3] = [0.004, 3e4, 0.03 if quant else 0.001]
        draw = lambda: rng.gamma(params[0], 1.0 / params[1], (rows + 1, P))
        special = [nan, inf, -inf, 0.0, -1.0, 5e-324, 1e300]
    elif name == "inversegamma":  # (m355_dists2.h: tables for alpha in [0.05, 1e4])
        params = (rng.uniform(1.0, 30.0, P), rng.uniform(0.2, 5.0, P))
        params[0][:3] = [0.02, 3e4, 0.04]
        draw = lambda: params[1] / rng.gamma(params[0], 1.0, (rows + 1, P))
        special = [nan, inf, -inf, 0.0, -1.0, 5e-324, 1e300]
    elif name == "beta":
        params = (rng.uniform(1.0, 9.0, P), rng.uniform(1.0, 20.0, P))
        if fn == "cdf":
            params[1][P // 2:] = rng.uniform(0.1, 1.0, P - P // 2)
        params[0][:3] = [0.004, 400.0, 0.3]
        params[1][:3] = [2.0, 300.0, 2.0]
        draw = lambda: rng.beta(params[0], params[1], (rows + 1, P))
        special = [nan, inf, -inf, 0.0, 1.0, -0.5, 1.5, 5e-324]
    elif name == "halfstudentt":
        params = (rng.uniform(1.0, 50.0, P), rng.uniform(0.1, 4.0, P))
        params[0][:3] = [0.5, 80.0, 1e6]
        draw = lambda: params[1] * np.abs(rng.standard_t(params[0], (rows + 1, P)))
        special = [nan, inf, -inf, 0.0, -1.0, 5e-324, 1e30]
    else:
        params = (rng.uniform(1.0, 50.0, P), rng.normal(0, 3, P), rng.uniform(0.1, 4.0, P))
        params[0][:3] = [0.5, 80.0, 1e6]
        draw = lambda: params[1] + params[2] * rng.standard_t(params[0], (rows + 1, P))
        special = [nan, inf, -inf, 1e30, -1e30]
    if quant:
        x = rng.uniform(0, 1, (rows + 1, P))
        x[1, :] = np.exp(rng.uniform(-20, 0, P))
        x[2, :] = 1 - np.exp(rng.uniform(-20, 0, P))
        special = [nan, 0.0, 1.0, -0.1, 1.1, 1e-300, 1e-12, 1 - 2.0**-53, 0.5]
    else:
        x = draw()
    for j, sv in enumerate(special):
        x[3 + j, :] = sv
    x = np.ascontiguousarray(x).ravel()[:n]
    xt = torch.from_numpy(x).to(gpu)
    keep = [torch.from_numpy(np.ascontiguousarray(p)).to(gpu) for p in params]
    wsb = torch.empty(int(L.m355_elemwise_workspace_bytes()), dtype=torch.uint8, device=gpu)  # k_coltab's tables live here

    def flat_call():
        out = torch.empty(n, dtype=torch.float64, device=gpu)
        ops = (Operand * (1 + len(params)))()
        ops[0].ptr, ops[0].div, ops[0].mod, ops[0].dtype = xt.data_ptr(), 1, n, F64
        for j, p in enumerate(keep, start=1):
            ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = p.data_ptr(), 0.0, 1, P, F64
        at = lambda k: ctypes.cast(ctypes.byref(ops, k * ctypes.sizeof(Operand)), ctypes.POINTER(Operand))
        _lib.check(L.m355_elemwise_ws(DIST[name], FN[fn], F64, n, at(0), len(params), at(1), out.data_ptr(),
                                      wsb.data_ptr(), wsb.numel(), torch.cuda.current_stream(gpu).cuda_stream))
        return out.cpu().numpy()

    old = L.m355_set_tab_min_n(2**62)
    try:
        exact = flat_call()
        L.m355_set_tab_min_n(1)
        tab = flat_call()
    finally:
        L.m355_set_tab_min_n(old)
    col = np.arange(n) % P
    pfull = [np.resize(p, n) for p in params]
    with np.errstate(all="ignore"):
        exp = R.TABLE[f"{name}_{fn}"](x, *pfull)
    assert np.array_equal(np.isnan(tab), np.isnan(exact))
    declined = col < (2 if (name, fn) in (("beta", "cdf"), ("beta", "isf")) else 3)  # (beta(0.3, 2): cdf and isf tabulated, logsf not)
    assert np.array_equal(tab[declined], exact[declined], equal_nan=True)
    assert np.array_equal(tab[rows * P:], exact[rows * P:], equal_nan=True)  # the partial row
    sp_rows = slice(3 * P, (3 + len(special)) * P)
    if not quant:
        assert np.array_equal(tab[sp_rows], exact[sp_rows], equal_nan=True)
    fin = np.isfinite(exact)
    if quant:
        extra = np.zeros(n)
        if name in ("studentt", "halfstudentt"):
            # harness._student_window (the reference's sqrt(1/x - 1) next to the median resp. next to 0), per element,
            # modelled at the table's value (the smooth one: the reference's own can be 0 where b rounds to 1)
            mu_, sg_ = (pfull[1], pfull[2]) if name == "studentt" else (0.0, pfull[1])
            with np.errstate(all="ignore"):
                tt = np.abs((tab - mu_) / sg_)
                delta = tt * tt / (pfull[0] + tt * tt)
                amp = np.where(delta > 0, 4 * 1.1e-16 / (2 * delta), 0.0) + 2e-16 * pfull[0]
                extra = np.where(np.isfinite(tt), amp * tt * sg_, 0.0)
            assert np.all(np.abs(tab[fin] - exact[fin]) <= 1e-12 * np.maximum(1.0, np.abs(exact[fin])) + extra[fin])
        else:
            with np.errstate(all="ignore"):
                lg = np.where(exact[fin] > 0, np.abs(np.log(np.where(exact[fin] > 0, exact[fin], 1.0))), 0.0)
            assert np.all(np.abs(tab[fin] - exact[fin]) <= (2e-13 + 3e-14 * lg) * np.abs(exact[fin]))
        with np.errstate(all="ignore"):
            ok = H.close(tab, exp) | (np.abs(tab - exp) <= 1e-12 * np.maximum(1.0, np.abs(exp)) + extra)
        ok |= H.ppf_unpinned(name, fn, x) | ~np.isfinite(exp) | (tab == exact)
    else:
        assert np.all(np.abs(tab[fin] - exact[fin]) <= 5e-13 * np.maximum(1.0, np.abs(exact[fin])))
        ok = H.close(tab, exp) | H.family_ok(name, fn, tab, exp) | (tab == exact)
    assert np.all(ok), (name, fn, x[~ok][:4], tab[~ok][:4], exp[~ok][:4])
    used = ~declined & (np.arange(n) < rows * P)
    assert np.mean(tab[used] == exact[used]) < 0.5  # the tables really were used
    if (name, fn) in (("gamma", "logcdf"), ("beta", "isf")):  # float32 in / out through the same kernel (generic loads)
        from distributions_amd._lib import F32

        with np.errstate(over="ignore"):  # (1e300 -> inf)
            x32 = torch.from_numpy(x.astype(np.float32)).to(gpu)

        def flat_call32():
            out = torch.empty(n, dtype=torch.float32, device=gpu)
            ops = (Operand * (1 + len(params)))()
            ops[0].ptr, ops[0].div, ops[0].mod, ops[0].dtype = x32.data_ptr(), 1, n, F32
            for j, p in enumerate(keep, start=1):
                ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = p.data_ptr(), 0.0, 1, P, F64
            at = lambda k: ctypes.cast(ctypes.byref(ops, k * ctypes.sizeof(Operand)), ctypes.POINTER(Operand))
            _lib.check(L.m355_elemwise_ws(DIST[name], FN[fn], F32, n, at(0), len(params), at(1), out.data_ptr(),
                                          wsb.data_ptr(), wsb.numel(), torch.cuda.current_stream(gpu).cuda_stream))
            return out.cpu().numpy().astype(np.float64)

        old = L.m355_set_tab_min_n(2**62)
        try:
            exact32 = flat_call32()
            L.m355_set_tab_min_n(1)
            tab32 = flat_call32()
        finally:
            L.m355_set_tab_min_n(old)
        assert np.array_equal(np.isnan(tab32), np.isnan(exact32))
        fin = np.isfinite(exact32)
        assert np.array_equal(np.isinf(tab32), np.isinf(exact32))
        assert np.all(np.abs(tab32[fin] - exact32[fin]) <= 1e-5 * np.maximum(1.0, np.abs(exact32[fin])))


def test_float32_arithmetic_bodies(gpu):
    """beta.logpdf on float32 data takes its two logarithms in float32 while the parameters are inside
    Dist<beta>::f32_ok (|alpha - 1|, |beta - 1| <= 8, |betaln| <= 12) and the float64 body outside: both within the
    float32 bar 1e-5 * max(1, |b|) of the oracle on the same float32 inputs, over the parameter box and at the edges
    (0, 1, out of support, denormal and near-1 inputs, NaN)."""
    from distributions_amd import beta

    rng = np.random.default_rng(61)
    n = 300_001
    worst = 0.0
    for a, b in [(2.0, 5.0), (0.1, 0.1), (0.5, 3.0), (1.0, 1.0), (9.0, 9.0), (9.0, 0.2), (1.0, 9.0), (4.5, 8.9), (0.11, 9.0),
                 (100.0, 100.0), (30.0, 2.0)]:  # the last two are outside the box: float64 body
        x = rng.beta(min(a, 20.0), min(b, 20.0), n).astype(np.float32)
        x[:12] = [0.0, 1.0, -0.5, 1.5, np.nan, 1e-40, 1.2e-38, 1 - 6e-8, 0.5, 6e-8, 1e-30, 0.999]
        x[12:2000] = np.exp(rng.uniform(-87, 0, 1988)).astype(np.float32)  # the whole float32 range of log x
        got = beta.logpdf(x, np.float32(a), np.float32(b))
        assert got.dtype == np.float32
        with np.errstate(all="ignore"):
            exp = R.beta_logpdf(x.astype(np.float64), float(np.float32(a)), float(np.float32(b)))
        ok = H.close(got.astype(np.float64), exp, 1e-5)
        assert np.all(ok), (a, b, x[~ok][:5], got[~ok][:5], exp[~ok][:5])
        with np.errstate(all="ignore"):
            fin = np.isfinite(exp)
            worst = max(worst, float(np.max(np.abs(got[fin] - exp[fin]) / np.maximum(1.0, np.abs(exp[fin])))))
    assert worst <= 1e-5


@pytest.mark.parametrize("P", [1024, 1000, 37, 8])
def test_sumlogp_periodic_product_form(gpu, P):
    """Fused sum-logp with row-periodic parameters: k_periodic keeps one product-form accumulator per column when the
    grid makes a thread's columns fixed (gridDim * 1024 a multiple of P) and every column's coefficients are within
    the product form's limit; totals against math.fsum of the oracle's terms (1e-12 * sum |term|), with exceptional
    elements (0, out of support, inf, NaN) and with one column above the limit (plain summation for the launch)."""
    from distributions_amd import beta, gamma, studentt

    rng = np.random.default_rng(71)
    rows = 3_000_000 // P + 3
    for m, name, draw, params in (
        (gamma, "gamma", lambda: rng.gamma(2.5, 1.0, (rows, P)), (rng.uniform(0.5, 9.0, P), rng.uniform(0.3, 3.0, P))),
        (beta, "beta", lambda: rng.beta(2, 5, (rows, P)), (rng.uniform(0.5, 9.0, P), 4.0)),
        (studentt, "studentt", lambda: 2 * rng.standard_t(5, (rows, P)), (rng.uniform(2.0, 30.0, P), 0.1, rng.uniform(0.5, 2.0, P))),
    ):
        x = draw()
        for extra in ([], [0.0], [-1.0, np.inf], [np.nan]):
            xx = x.copy()
            for j, e in enumerate(extra):
                xx[(j + 1) * 17, (j * 5) % P] = e
            with np.errstate(all="ignore"):
                terms = R.TABLE[f"{name}_logpdf"](xx, *params)
                exp = float(np.sum(terms))
            got = m.sum_logpdf(xx, *params)
            if math.isfinite(exp):
                assert abs(got - math.fsum(terms.ravel())) <= 1e-12 * float(np.abs(terms).sum()), (name, P, extra, got, exp)
            else:
                assert (math.isnan(exp) and math.isnan(got)) or got == exp, (name, P, extra, got, exp)
        if name == "gamma":  # one column beyond kSumAccMaxCoef: the launch falls back to summing logpdf()
            a2 = params[0].copy()
            a2[P // 2] = 5000.0
            terms = R.gamma_logpdf(x, a2, params[1])
            assert abs(m.sum_logpdf(x, a2, params[1]) - math.fsum(terms.ravel())) <= 1e-12 * float(np.abs(terms).sum())
