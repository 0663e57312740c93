"""Pins the CPU oracle (oracle/restatement.py) against everything the reference's own tests hold
for the hot path: the SciPy comparisons of tests/helper_scipy.py on the reference's parameter
grids, the multivariate cases of tests/test_mvnormal.py / test_mvstudentt.py /
test_matrixnormal.py, and the integer-logic cases of tests/test_categorical.py /
test_multinomial.py (all in /root/reference/tests; values restated here as data).  CPU only."""

import numpy as np
import pytest
from numpy.testing import assert_allclose
from scipy import stats

from oracle import restatement as R
from tests import harness as H


@pytest.mark.parametrize("name, sp_dist, params, sp_params, support, kw", H.grids(), ids=H.grid_ids())
def test_oracle_vs_scipy_harness_a(name, sp_dist, params, sp_params, support, kw):
    H.run_distribution_checks(H.oracle_namespace(name), sp_dist, params, sp_params, support, **kw)


MVN_CASES = [
    (np.array([0.0, 0.0]), np.eye(2)),
    (np.array([1.0, -1.0]), np.array([[2.0, 0.0], [0.0, 0.5]])),
    (np.array([0.0, 0.0, 0.0]), np.eye(3)),
    (np.array([1.0, 2.0, 3.0]), np.array([[1.0, 0.5, 0.2], [0.5, 2.0, -0.3], [0.2, -0.3, 0.8]])),
]
MVT_NU = [5.0, 3.0, 10.0, 2.5]


@pytest.mark.parametrize("mu, cov", MVN_CASES)
def test_oracle_mvnormal(mu, cov):
    d = stats.multivariate_normal(mean=mu, cov=cov)
    assert_allclose(R.mvnormal_logpdf(mu, mu, cov), d.logpdf(mu), rtol=1e-3)
    xs = d.rvs(size=10, random_state=814)
    assert_allclose(R.mvnormal_logpdf(xs, mu, cov), d.logpdf(xs), rtol=1e-5)
    assert R.mvnormal_logpdf(mu, mu, cov).shape == ()
    with pytest.raises(ValueError):
        R.mvnormal_logpdf(np.float64(1.0), mu, cov)


@pytest.mark.parametrize("nu, case", list(zip(MVT_NU, MVN_CASES)))
def test_oracle_mvstudentt(nu, case):
    mu, cov = case
    d = stats.multivariate_t(df=nu, loc=mu, shape=cov)
    assert_allclose(R.mvstudentt_logpdf(mu, nu, mu, cov), d.logpdf(mu), rtol=1e-3)
    xs = d.rvs(size=10, random_state=814)
    assert_allclose(R.mvstudentt_logpdf(xs, nu, mu, cov), d.logpdf(xs), rtol=1e-5)


MATN_CASES = [
    (np.array([[1.0, 2.0], [3.0, 4.0]]), np.eye(2), np.eye(2)),
    (
        np.array([[1.0, 0.0, -1.0], [2.0, 1.0, 0.0]]),
        np.array([[2.0, 0.5], [0.5, 1.0]]),
        np.array([[1.0, 0.3, 0.1], [0.3, 2.0, 0.4], [0.1, 0.4, 1.5]]),
    ),
    (
        np.zeros((3, 2)),
        np.array([[3.0, 1.0, 0.5], [1.0, 2.0, 0.3], [0.5, 0.3, 1.5]]),
        np.array([[1.0, 0.7], [0.7, 2.0]]),
    ),
]


@pytest.mark.parametrize("M, U, V", MATN_CASES)
def test_oracle_matrixnormal(M, U, V):
    d = stats.matrix_normal(mean=M, rowcov=U, colcov=V)
    X = M + 0.1 * np.ones_like(M)
    assert_allclose(R.matrixnormal_logpdf(X, M, U, V), d.logpdf(X), rtol=1e-5)


def test_oracle_matrixnormal_batched():
    M0 = np.array([[1.0, 2.0], [3.0, 4.0]])
    U0 = np.array([[2.0, 0.5], [0.5, 1.0]])
    V0 = np.eye(2)
    Xs = np.array([M0 + 0.1, M0 - 0.1, M0 + 0.5])
    d = stats.matrix_normal(mean=M0, rowcov=U0, colcov=V0)
    got = R.matrixnormal_logpdf(Xs, M0, U0, V0)
    assert got.shape == (3,)
    assert_allclose(got, [d.logpdf(x) for x in Xs], rtol=1e-5)
    assert R.matrixnormal_logpdf(M0 + 0.1, M0, U0, V0).shape == ()
    rowcovs = np.array([U0, 2 * np.eye(2)])
    got = R.matrixnormal_logpdf(M0 + 0.1, M0, rowcovs, V0)
    assert got.shape == (2,)
    for i in range(2):
        assert_allclose(got[i], stats.matrix_normal(mean=M0, rowcov=rowcovs[i], colcov=V0).logpdf(M0 + 0.1),
                        rtol=1e-5)


@pytest.mark.parametrize(
    "probs",
    [[0.5, 0.5], [0.25, 0.25, 0.25, 0.25], [0.1, 0.2, 0.3, 0.4], [0.7, 0.2, 0.1], [0.01, 0.01, 0.98]],
)
def test_oracle_categorical(probs):
    k = len(probs)
    pr = np.asarray(probs) / np.sum(probs)
    d = stats.rv_discrete(values=(np.arange(k), pr))
    xs = np.array([-2, -1, 0, 1, k - 1, k, k + 1], dtype="float64")
    with np.errstate(all="ignore"):
        assert_allclose(R.categorical_logpdf(xs, probs), d.logpmf(xs), rtol=1e-6)
        assert_allclose(R.categorical_cdf(xs, probs), d.cdf(xs), rtol=1e-6)
    qs = np.array([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0])
    assert_allclose(R.categorical_ppf(qs, probs), d.ppf(qs))
    # non-integer and fractional points (categorical.py:69-71): never in support
    assert np.all(np.isneginf(R.categorical_logpdf(np.array([0.5, 1.25, -0.5]), probs)))


@pytest.mark.parametrize(
    "n, p",
    [(10, [0.5, 0.5]), (20, [0.2, 0.3, 0.5]), (15, [0.25] * 4), (30, [0.1, 0.2, 0.3, 0.4])],
)
def test_oracle_multinomial(n, p):
    p = np.asarray(p)
    d = stats.multinomial(n=n, p=p)
    x = np.round(d.mean()).astype(int)
    x[-1] = n - x[:-1].sum()
    assert_allclose(R.multinomial_logpdf(x, n, p), d.logpmf(x), rtol=1e-5)


def test_oracle_multinomial_constraints():
    p = np.array([0.5, 0.5])
    assert R.multinomial_logpdf(np.array([-1, 11]), 10, p) == -np.inf
    assert R.multinomial_logpdf(np.array([5, 6]), 10, p) == -np.inf


@pytest.mark.parametrize("name, params", H.empirical_discrete_cases(),
                         ids=[f"{n}-{'-'.join(map(str, p))}" for n, p in H.empirical_discrete_cases()])
def test_oracle_discrete_without_scipy(name, params):
    """discreteweibull and the zero-inflated families have no SciPy counterpart (the reference tests
    them against its own samples, tests/test_zi_poisson.py:11-50 etc.): the restatement is pinned on
    the defining mixture / closed form written with scipy.stats, and ppf by brute force."""
    ks = np.arange(0, 400, dtype=np.float64)
    if name == "discreteweibull":
        q, b = params
        pmf = q ** (ks ** b) - q ** ((ks + 1) ** b)
        cdf = 1 - q ** ((ks + 1) ** b)
    else:
        psi = params[0]
        base = {"zi_poisson": lambda: stats.poisson(params[1]),
                "zi_binomial": lambda: stats.binom(int(params[1]), params[2]),
                "zi_negativebinomial": lambda: stats.nbinom(params[1], params[2])}[name]()
        pmf = psi * base.pmf(ks) + (1 - psi) * (ks == 0)
        cdf = (1 - psi) + psi * base.cdf(ks)
    ns = H.oracle_namespace(name)
    with np.errstate(all="ignore"):
        assert_allclose(ns.pdf(ks, *params), pmf, rtol=1e-9, atol=1e-300)
        assert_allclose(np.exp(ns.logpdf(ks, *params)), pmf, rtol=1e-9, atol=1e-300)
        assert_allclose(ns.cdf(ks, *params), cdf, rtol=1e-9)
        # quirk kept: discreteweibull.cdf (discreteweibull.py:49-50), poisson.cdf (poisson.py:49-50) and
        # binomial.cdf do not floor x; negativebinomial.cdf (negativebinomial.py:46-48) does
        if name == "zi_negativebinomial":
            assert_allclose(ns.cdf(ks + 0.5, *params), cdf, rtol=1e-9)
        assert_allclose(ns.sf(ks, *params), 1 - cdf, rtol=1e-6, atol=1e-14)
        assert_allclose(np.exp(ns.logcdf(ks, *params)), cdf, rtol=1e-9)
        assert_allclose(np.exp(ns.logsf(ks, *params)), 1 - cdf, rtol=1e-6, atol=1e-14)
        assert np.all(ns.pdf(np.array([-1.0, -np.inf]), *params) == 0)
        assert np.all(ns.cdf(np.array([-1.0, -np.inf]), *params) == 0) and ns.cdf(np.inf, *params) == 1
    qs = np.random.default_rng(4).uniform(0.001, 0.999, 500)
    brute = np.searchsorted(ns.cdf(ks, *params), qs, side="left").astype(np.float64)
    got = ns.ppf(qs, *params)
    # find_ppf_discrete (optimization.py:102-129) cycles between the bounds for some q of the
    # zero-inflated families (e.g. 20 -> 0 -> 1 -> 20 for zi_binomial(0.5, 20, 0.5)) and then returns
    # wherever iteration 100 stands: the restatement keeps that, so only converged points are compared
    conv = (ns.cdf(got, *params) >= qs - 1e-7) & (ns.cdf(got - 1, *params) < qs)
    assert conv.mean() > 0.3
    assert np.array_equal(got[conv], brute[conv])
    assert np.array_equal(ns.isf(qs, *params), ns.ppf(1 - qs, *params))
    edge = ns.ppf(np.array([-0.1, 1.1]), *params)
    assert np.all(np.isnan(edge))


@pytest.mark.parametrize("h, z", [(1.0, 0.0), (2.0, 2.0), (0.5, 1.0)])  # tests/test_polyagamma.py:13-17
def test_oracle_polyagamma(h, z):
    """polyagamma has no SciPy counterpart (the reference tests it against its own samples at rtol 3e-2 .. 3e-1,
    tests/test_polyagamma.py:34-57): the restated density is pinned on what defines PG(h, z) -- total mass 1, the
    mean h/(2z) tanh(z/2), and the Laplace transform E exp(-t X) = (cosh(z/2) / cosh(sqrt((z^2/2 + t)/2)))^h -- and
    the restated cdf (a 500-point trapezoid, polyagamma.py:124-134) on the integral of that density."""
    from scipy.integrate import quad

    pdf = lambda x: float(R.TABLE["polyagamma_pdf"](x, h, z))  # noqa: E731
    assert abs(quad(pdf, 0, np.inf, limit=200)[0] - 1.0) < 1e-10
    mean = h / 4.0 if z == 0 else h / (2 * z) * np.tanh(z / 2)
    assert abs(quad(lambda x: x * pdf(x), 0, np.inf, limit=200)[0] - mean) < 1e-10
    for t in (0.3, 1.3, 7.0):
        exact = (np.cosh(z / 2) / np.cosh(np.sqrt((z * z / 2 + t) / 2))) ** h
        assert abs(quad(lambda x: np.exp(-t * x) * pdf(x), 0, np.inf, limit=200)[0] - exact) < 1e-10
    xs = np.array([0.05, 0.2, 0.5, 1.0])
    assert_allclose(R.TABLE["polyagamma_cdf"](xs, h, z), [quad(pdf, 0, x)[0] for x in xs], rtol=1e-4)
    ext = np.array([0.0, np.inf, -1.0, -2.0])  # tests/test_polyagamma.py:60-66
    assert_allclose(R.TABLE["polyagamma_cdf"](ext, h, z), [0.0, 1.0, 0.0, 0.0])
    q = np.array([0.05, 0.35, 0.65, 0.95])
    assert_allclose(R.TABLE["polyagamma_cdf"](R.TABLE["polyagamma_ppf"](q, h, z), h, z), q, atol=1e-7)
    assert_allclose(R.TABLE["polyagamma_isf"](q, h, z), R.TABLE["polyagamma_ppf"](1 - q, h, z))


def test_oracle_edge_table_survey_appendix_c():
    """Edge values recorded in SURVEY.md Appendix C (literal float64 restatement)."""
    inf = np.inf
    x = np.array([-inf, -40, -1, 0, 1, 40, inf])
    assert_allclose(R.normal_logpdf(x, 0, 1), [-inf, -800.918938533, -1.418938533, -0.918938533,
                                               -1.418938533, -800.918938533, -inf], rtol=1e-9)
    got = R.normal_ppf(np.array([-0.1, 0, 1e-300, 0.5, 1 - 1e-16, 1, 1.1]), 0, 1)
    assert np.isnan(got[0]) and np.isnan(got[-1]) and got[1] == -inf and got[2] == -inf and got[5] == inf
    assert_allclose(got[3:5], [0, 8.20954], rtol=1e-5)
    assert_allclose(R.poisson_logpdf(np.array([-2, -1, -0.5, 0, 1, 2.5]), 2.0),
                    [-inf, -inf, -2.91894, -2, -1.30685, -1.46811], rtol=1e-5)
    assert_allclose(R.binomial_logpdf(np.array([-1, 0, 3, 3.5, 10, 11, inf]), 10, 0.3),
                    [-inf, -3.56675, -1.32115, -1.41598, -12.0397, -inf, -inf], rtol=1e-5)
    # quirk Q2: gamma(alpha=1) at 0 and +inf
    assert_allclose(R.gamma_logpdf(np.array([0.0, inf]), 1.0, 1.0), [0.0, -inf])


def test_oracle_c_leg_matches_restatement():
    """oracle/elemwise_c.c (bench.py's cpu_baseline) against oracle/restatement.py."""
    from oracle import c_leg

    L = c_leg.load()
    rng = np.random.default_rng(3)
    for name, p, x in (("normal", (0.3, 1.7), rng.normal(0, 1, 5000)),
                       ("gamma", (2.5, 1.3), rng.gamma(2.5, 1 / 1.3, 5000)),
                       ("gamma", (1.0, 1.0), np.array([0.0, 1.0, np.inf, -1.0])),
                       ("beta", (2.0, 5.0), np.concatenate([rng.beta(2, 5, 5000), [0.0, 1.0, -0.5, 1.5]]))):
        with np.errstate(all="ignore"):
            exp = R.TABLE[f"{name}_logpdf"](x, *p)
            got = c_leg.logpdf(L, name, x, *p)
        assert np.all(H.close(got, exp)), name
