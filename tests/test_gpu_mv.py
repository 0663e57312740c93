"""GPU parity tests of mvnormal / mvstudentt / matrixnormal logpdf against the oracle (SciPy LAPACK
restatement) and against scipy.stats on the reference's own cases (tests/test_mvnormal.py:12-37,
tests/test_mvstudentt.py:11-38, tests/test_matrixnormal.py:13-55,121-145,267-276 of the reference)."""

import numpy as np
import pytest
from numpy.testing import assert_allclose
from scipy import stats

from oracle import restatement as R
from tests import harness as H
from tests.test_oracle_pinning import MATN_CASES, MVN_CASES, MVT_NU

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mu, cov", MVN_CASES)
def test_mvnormal_reference_cases(gpu, mu, cov):
    from distributions_amd import mvnormal

    d = stats.multivariate_normal(mean=mu, cov=cov)
    r = mvnormal.logpdf(mu, mu, cov)
    assert r.shape == ()
    assert_allclose(r, d.logpdf(mu), rtol=1e-3)
    xs = d.rvs(size=10, random_state=814)
    assert_allclose(mvnormal.logpdf(xs, mu, cov), d.logpdf(xs), rtol=1e-5)
    assert np.all(H.close(mvnormal.logpdf(xs, mu, cov), R.mvnormal_logpdf(xs, mu, cov)))
    assert_allclose(mvnormal.pdf(mu, mu, cov), d.pdf(mu), rtol=1e-5)
    with pytest.raises(ValueError, match="scalar"):
        mvnormal.logpdf(np.float64(1.0), mu, cov)


@pytest.mark.parametrize("nu, case", list(zip(MVT_NU, MVN_CASES)))
def test_mvstudentt_reference_cases(gpu, nu, case):
    from distributions_amd import mvstudentt

    mu, cov = case
    d = stats.multivariate_t(df=nu, loc=mu, shape=cov)
    assert_allclose(mvstudentt.logpdf(mu, nu, mu, cov), d.logpdf(mu), rtol=1e-3)
    xs = d.rvs(size=10, random_state=814)
    assert_allclose(mvstudentt.logpdf(xs, nu, mu, cov), d.logpdf(xs), rtol=1e-5)
    assert np.all(H.close(mvstudentt.logpdf(xs, nu, mu, cov), R.mvstudentt_logpdf(xs, nu, mu, cov)))


def config4_inputs(n, d=64):
    A = np.random.default_rng(6).normal(size=(d, d))
    cov = A @ A.T / d + np.eye(d)
    mu = np.random.default_rng(7).normal(size=d)
    z = np.random.default_rng(8).normal(size=(n, d))
    return mu + z @ np.linalg.cholesky(cov).T, mu, cov


def test_mvnormal_config4_shared_cov(gpu):
    """SURVEY.md 8(d) config 4 (seeds 6/7/8), d = 64: 1e-12 vs the restatement, 1e-10 vs scipy.stats."""
    from distributions_amd import mvnormal, mvstudentt

    X, mu, cov = config4_inputs(20_001)
    got = mvnormal.logpdf(X, mu, cov)
    assert got.shape == (20_001,)
    assert np.all(H.close(got, R.mvnormal_logpdf(X, mu, cov)))
    assert np.all(H.close(got, stats.multivariate_normal(mu, cov).logpdf(X), 1e-10))
    assert np.all(H.close(mvstudentt.logpdf(X, 7.5, mu, cov), R.mvstudentt_logpdf(X, 7.5, mu, cov)))
    # other dimensions incl. the LDS limit d = 128, leading batch axes of the value
    for d in (1, 5, 16, 33, 128):
        Xd, mud, covd = config4_inputs(300, d)
        assert np.all(H.close(mvnormal.logpdf(Xd, mud, covd), R.mvnormal_logpdf(Xd, mud, covd))), d
    X3 = X[:600].reshape(20, 30, 64)
    assert np.all(H.close(mvnormal.logpdf(X3, mu, cov), R.mvnormal_logpdf(X3, mu, cov)))
    assert mvnormal.logpdf(np.zeros((0, 64)), mu, cov).shape == (0,)


def test_mvnormal_batched_cov_and_mu(gpu):
    from distributions_amd import mvnormal

    rng = np.random.default_rng(20)
    B, d = 37, 12
    A = rng.normal(size=(B, d, d))
    covs = A @ np.swapaxes(A, -1, -2) / d + np.eye(d)
    mus = rng.normal(size=(B, d))
    X = rng.normal(size=(B, d))
    assert np.all(H.close(mvnormal.logpdf(X, mus, covs), R.mvnormal_logpdf(X, mus, covs)))  # one cov per row
    X2 = rng.normal(size=(B, 9, d))  # cov batch (B,1) against values (B,9): "outer" index
    got = mvnormal.logpdf(X2, mus[:, None, :], covs[:, None])
    assert got.shape == (B, 9)
    assert np.all(H.close(got, R.mvnormal_logpdf(X2, mus[:, None, :], covs[:, None])))
    X3 = rng.normal(size=(5, B, d))  # cov batch (B,) against values (5,B): "inner" index
    assert np.all(H.close(mvnormal.logpdf(X3, mus, covs), R.mvnormal_logpdf(X3, mus, covs)))


def test_mvnormal_not_positive_definite_gives_nan(gpu):
    """Quirk Q5: posdef is ignored (mvnormal.py:81); a non-PD covariance yields NaN, not -inf."""
    from distributions_amd import mvnormal

    cov = np.array([[1.0, 2.0], [2.0, 1.0]])
    assert np.isnan(mvnormal.logpdf(np.array([[0.1, 0.2]]), np.zeros(2), cov)).all()


@pytest.mark.parametrize("M, U, V", MATN_CASES)
def test_matrixnormal_reference_cases(gpu, M, U, V):
    from distributions_amd import matrixnormal

    X = M + 0.1 * np.ones_like(M)
    got = matrixnormal.logpdf(X, M, U, V)
    assert got.shape == ()
    assert_allclose(got, stats.matrix_normal(mean=M, rowcov=U, colcov=V).logpdf(X), rtol=1e-5)
    assert np.all(H.close(got, R.matrixnormal_logpdf(X, M, U, V)))


def test_matrixnormal_batched(gpu):
    from distributions_amd import matrixnormal

    M0 = np.array([[1.0, 2.0], [3.0, 4.0]])
    U0 = np.array([[2.0, 0.5], [0.5, 1.0]])
    V0 = np.eye(2)
    Xs = np.array([M0 + 0.1, M0 - 0.1, M0 + 0.5])
    got = matrixnormal.logpdf(Xs, M0, U0, V0)
    assert got.shape == (3,)
    assert np.all(H.close(got, R.matrixnormal_logpdf(Xs, M0, U0, V0)))
    rowcovs = np.array([U0, 2 * np.eye(2)])
    got = matrixnormal.logpdf(M0 + 0.1, M0, rowcovs, V0)
    assert got.shape == (2,)
    assert np.all(H.close(got, R.matrixnormal_logpdf(M0 + 0.1, M0, rowcovs, V0)))
    rng = np.random.default_rng(21)
    m, n, B = 9, 14, 50
    A, C = rng.normal(size=(m, m)), rng.normal(size=(B, n, n))
    U = A @ A.T / m + np.eye(m)
    V = C @ np.swapaxes(C, -1, -2) / n + np.eye(n)
    Mu = rng.normal(size=(m, n))
    X = rng.normal(size=(B, m, n))
    assert np.all(H.close(matrixnormal.logpdf(X, Mu, U, V), R.matrixnormal_logpdf(X, Mu, U, V)))


def test_one_dim_value_against_batched_parameters(gpu):
    """mvnormal.py:34-35: a 1-d value returns quaddist[0]; with batched mu / cov that is the quadratic form of the
    first batch entry broadcast against every entry's logdet -- the reference's behaviour, reproduced (the oracle
    restates it line by line)."""
    from distributions_amd import mvnormal, mvstudentt

    rng = np.random.default_rng(77)
    d, B = 5, 7
    A = rng.normal(size=(B, d, d))
    cov = A @ np.swapaxes(A, 1, 2) / d + np.eye(d)
    mu = rng.normal(size=(B, d))
    x = rng.normal(size=d)
    for m_, c_ in ((mu[0], cov), (mu, cov[0]), (mu, cov)):
        got = mvnormal.logpdf(x, m_, c_)
        exp = R.mvnormal_logpdf(x, m_, c_)
        assert got.shape == exp.shape == ((B,) if np.ndim(c_) == 3 else ())  # logdet's shape against quaddist[0]'s
        assert np.all(H.close(got, exp)), (got, exp)
        got = mvstudentt.logpdf(x, 4.5, m_, c_)
        assert np.all(H.close(got, R.mvstudentt_logpdf(x, 4.5, m_, c_)))
    cov2 = np.stack([cov[:6].reshape(2, 3, d, d)])[0]
    got = mvnormal.logpdf(x, mu[0], cov2)
    assert got.shape == (2, 3) and np.all(H.close(got, R.mvnormal_logpdf(x, mu[0], cov2)))


@pytest.mark.parametrize("d", [64, 63, 48, 33, 32, 17, 16, 7, 1])
def test_one_covariance_per_observation_fused(gpu, d):
    """SURVEY.md 8(d) config 4, variant B: cov carries the batch axis (mvnormal.py:27) -- one wavefront factors each
    matrix in registers and solves its observation (k_mvn_tiles; d covers every tile count, the full-tile load path at
    16 / 32 / 48 / 64 and the clamped one next to them).  1e-12 against the restatement, shared and per-row means, both
    kinds; a non-positive-definite matrix gives NaN for its row only (quirk Q5); only the lower triangle of a
    covariance is read (numpy.linalg.cholesky's contract, mvnormal.py:9-14)."""
    from distributions_amd import mvnormal, mvstudentt

    rng = np.random.default_rng(91 + d)
    n = 1500
    A = rng.normal(size=(n, d, d))
    cov = A @ np.swapaxes(A, 1, 2) / d + np.eye(d)
    mu = rng.normal(size=d)
    mu_rows = rng.normal(size=(n, d))
    X = rng.normal(size=(n, d)) * 1.5
    for m_ in (mu, mu_rows):
        got = mvnormal.logpdf(X, m_, cov)
        assert got.shape == (n,)
        assert np.all(H.close(got, R.mvnormal_logpdf(X, m_, cov)))
        assert np.all(H.close(mvstudentt.logpdf(X, 6.5, m_, cov), R.mvstudentt_logpdf(X, 6.5, m_, cov)))
    if d > 1:
        bad = cov.copy()
        bad[5] = -np.eye(d)
        got = mvnormal.logpdf(X, mu, bad)
        assert np.isnan(got[5]) and np.all(H.close(np.delete(got, 5), np.delete(R.mvnormal_logpdf(X, mu, cov), 5)))
        semi = cov.copy()
        semi[7, :, -1] = 0.0  # a zero last pivot: log(0) - and 0/0 in the solve
        semi[7, -1, :] = 0.0
        got = mvnormal.logpdf(X, mu, semi)
        assert not np.isfinite(got[7]) and np.all(H.close(np.delete(got, 7), np.delete(R.mvnormal_logpdf(X, mu, cov), 7)))
        junk = cov + np.triu(rng.normal(size=(n, d, d)) * 50.0, 1)  # strictly upper part overwritten
        assert np.array_equal(mvnormal.logpdf(X, mu, junk), mvnormal.logpdf(X, mu, cov))
