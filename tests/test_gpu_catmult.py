"""GPU parity tests of categorical / multinomial (integer index and support logic, SURVEY.md rows
a31-a32) against the oracle and against SciPy on the reference's own cases
(tests/test_categorical.py:24-161, tests/test_multinomial.py:11-31,104-117 of the reference).
Decisions (in support / out of support, the gathered index, the ppf category) must be bit-exact;
gathered / summed float values are held to 1e-12."""

import numpy as np
import pytest
from numpy.testing import assert_allclose
from scipy import stats

from oracle import restatement as R
from tests import harness as H

pytestmark = pytest.mark.gpu

PROBS = [[0.5, 0.5], [0.25, 0.25, 0.25, 0.25], [0.1, 0.2, 0.3, 0.4], [0.7, 0.2, 0.1], [0.01, 0.01, 0.98]]
CAT_FNS = ["pdf", "logpdf", "cdf", "logcdf", "sf", "logsf"]


@pytest.mark.parametrize("probs", PROBS)
def test_categorical_reference_cases(gpu, probs):
    from distributions_amd import categorical

    k = len(probs)
    pr = np.asarray(probs) / np.sum(probs)
    d = stats.rv_discrete(values=(np.arange(k), pr))
    xs = np.array([-2, -1, 0, 1, k - 1, k, k + 1], dtype="float64")
    with np.errstate(all="ignore"):
        assert_allclose(categorical.pdf(xs, probs), d.pmf(xs), rtol=1e-6)
        assert_allclose(categorical.logpdf(xs, probs), d.logpmf(xs), rtol=1e-6)
        assert_allclose(categorical.cdf(xs, probs), d.cdf(xs), rtol=1e-6)
        assert_allclose(categorical.logcdf(xs, probs), d.logcdf(xs), rtol=1e-6)
        assert_allclose(categorical.sf(xs, probs), d.sf(xs), rtol=1e-6, atol=1e-15)
        qs = np.array([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0])
        assert_allclose(categorical.ppf(qs, probs), d.ppf(qs))
        assert_allclose(categorical.isf(qs, probs), d.isf(qs))
        # the oracle, every function, float and int64 observations, fractional / out-of-range points
        pts = np.array([-2, -1, -0.5, 0, 0.5, 1, 1.25, k - 1, k - 0.5, k, k + 1, np.inf, -np.inf], dtype="float64")
        for fn in CAT_FNS:
            got, exp = getattr(categorical, fn)(pts, probs), R.TABLE[f"categorical_{fn}"](pts, probs)
            assert np.array_equal(np.isinf(got), np.isinf(exp)) and np.all(H.close(got, exp)), fn
            ip = np.arange(-3, k + 3, dtype=np.int64)
            got, exp = getattr(categorical, fn)(ip, probs), R.TABLE[f"categorical_{fn}"](ip.astype(float), probs)
            assert np.array_equal(np.isinf(got), np.isinf(exp)) and np.all(H.close(got, exp)), fn
        qq = np.array([-0.1, 0.0, 1e-300, 0.3, 0.5, 1 - 1e-16, 1.0, 1.1, np.nan])
        assert np.array_equal(categorical.ppf(qq, probs), R.categorical_ppf(qq, probs), equal_nan=True)
        assert np.array_equal(categorical.isf(qq, probs), R.categorical_isf(qq, probs), equal_nan=True)
    assert np.isneginf(categorical.logpdf(np.nan, probs)) and categorical.logpdf(1, probs).shape == ()


def test_categorical_large_bit_exact(gpu):
    """1e6 observations over k = 1000 (LDS tables) and k = 5000 (global tables): the gathered
    category and the ppf category are bit-exact against the oracle."""
    from distributions_amd import categorical

    rng = np.random.default_rng(30)
    for k in (1000, 5000):
        p = rng.dirichlet(np.ones(k)) * 3.7  # un-normalised on purpose
        x = rng.integers(-5, k + 5, 1_000_003)
        exp = R.categorical_logpdf(x.astype(float), p)
        got = categorical.logpdf(x, p)
        assert np.array_equal(np.isneginf(got), np.isneginf(exp))
        assert np.all(H.close(got, exp))
        xf = x.astype(float) + rng.choice([0.0, 0.0, 0.5], x.size)
        got, exp = categorical.cdf(xf, p), R.categorical_cdf(xf, p)
        assert np.all(H.close(got, exp))
        q = rng.uniform(0, 1, 200_001)
        assert np.array_equal(categorical.ppf(q, p), R.categorical_ppf(q, p))
        assert categorical.logpdf(x[:77].reshape(11, 7), p).shape == (11, 7)
    with pytest.raises(ValueError, match="probability vector"):
        categorical.logpdf(1.0, np.ones((2, 3)))


def test_categorical_rvs_chi2(gpu):
    from distributions_amd import categorical

    p = np.array([0.1, 0.2, 0.3, 0.4]) * 2.0
    n = 400_000
    draws = categorical.rvs(p, size=n, random_state=7)
    assert draws.dtype == np.int64 and draws.min() >= 0 and draws.max() <= 3
    obs = np.bincount(draws, minlength=4)
    chi2 = float(np.sum((obs - n * p / p.sum()) ** 2 / (n * p / p.sum())))
    assert stats.chi2.sf(chi2, 3) > 1e-4
    again = categorical.rvs(p, size=n, random_state=7)
    assert np.array_equal(draws, again)
    halves = np.concatenate([categorical.rvs(p, size=n // 2, random_state=(7, 0)),
                             categorical.rvs(p, size=n // 2, random_state=(7, n // 2))])
    assert np.array_equal(draws, halves)  # partition invariance (counter = global index)


MULT_CASES = [(10, [0.5, 0.5]), (20, [0.2, 0.3, 0.5]), (15, [0.25] * 4), (30, [0.1, 0.2, 0.3, 0.4])]


@pytest.mark.parametrize("n, p", MULT_CASES)
def test_multinomial_reference_cases(gpu, n, p):
    from distributions_amd import multinomial

    p = np.asarray(p)
    d = stats.multinomial(n=n, p=p)
    x = np.round(d.mean()).astype(int)
    x[-1] = n - x[:-1].sum()
    got = multinomial.logpdf(x, n, p)
    assert got.shape == ()
    assert_allclose(got, d.logpmf(x), rtol=1e-5)
    assert_allclose(multinomial.pdf(x, n, p), d.pmf(x), rtol=1e-5)
    assert np.all(H.close(got, R.multinomial_logpdf(x, n, p)))


def test_multinomial_constraints_and_batches(gpu):
    from distributions_amd import multinomial

    p2 = np.array([0.5, 0.5])
    assert multinomial.logpdf(np.array([-1, 11]), 10, p2) == -np.inf  # tests/test_multinomial.py:111-113
    assert multinomial.logpdf(np.array([5, 6]), 10, p2) == -np.inf    # tests/test_multinomial.py:115-117
    rng = np.random.default_rng(31)
    for k in (1, 2, 3, 7, 16, 33, 64, 100, 257):
        p = rng.dirichlet(np.ones(k))
        rows = 5003
        x = rng.multinomial(40, p, size=rows)
        bad = rng.random(rows) < 0.2  # break the sum on a fifth of the rows, a negative count on some
        x[bad, 0] += 1
        neg = rng.random(rows) < 0.05
        x[neg, -1] = -1
        exp = R.multinomial_logpdf(x, 40, p)
        for xin in (x, x.astype(np.float64)):
            got = multinomial.logpdf(xin, 40, p)
            assert got.shape == (rows,)
            assert np.array_equal(np.isneginf(got), np.isneginf(exp)), k  # support decisions: exact
            assert np.all(H.close(got, exp)), k
    # per-row n and per-row p, leading batch axes
    k, rows = 5, 600
    P = rng.dirichlet(np.ones(k), size=rows)
    N = rng.integers(1, 60, rows)
    X = np.stack([rng.multinomial(N[i], P[i]) for i in range(rows)])
    got = multinomial.logpdf(X.reshape(20, 30, k), N.reshape(20, 30), P.reshape(20, 30, k))
    assert got.shape == (20, 30)
    assert np.all(H.close(got.ravel(), R.multinomial_logpdf(X, N, P)))
    assert multinomial.logpdf(np.zeros((0, 4)), 3, np.full(4, 0.25)).shape == (0,)


@pytest.mark.parametrize("n, p", MULT_CASES)
def test_multinomial_rvs(gpu, n, p):
    """tests/test_multinomial.py:69-90 of the reference: rows sum to n, sample mean / variance."""
    from distributions_amd import multinomial

    p = np.asarray(p)
    s = multinomial.rvs(n, p, size=20_000, random_state=3)
    assert s.dtype == np.int64 and s.shape == (20_000, len(p))
    assert np.all(s.sum(axis=1) == n) and s.min() >= 0
    assert_allclose(s.mean(axis=0), n * p, rtol=0.02)
    assert_allclose(s.var(axis=0), n * p * (1 - p), rtol=0.06)
    for j in range(len(p)):  # marginals are Binomial(n, p_j): chi-square on the counts
        obs = np.bincount(s[:, j], minlength=n + 1).astype(float)
        expc = stats.binom.pmf(np.arange(n + 1), n, p[j]) * s.shape[0]
        keep = expc >= 5
        o = np.append(obs[keep], obs[~keep].sum())
        e = np.append(expc[keep], expc[~keep].sum())
        m = e > 0
        chi2 = float(np.sum((o[m] - e[m]) ** 2 / e[m]))
        assert stats.chi2.sf(chi2, int(m.sum()) - 1) > 1e-4, (j, chi2)
