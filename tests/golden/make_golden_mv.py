"""Generates tests/golden/multivariate.npz: inputs and expected outputs for the multivariate,
categorical and multinomial rows of the path.  Expected values are
`scipy.stats` results on the reference's own test cases (tests/test_mvnormal.py:12-37,
test_mvstudentt.py:11-38, test_matrixnormal.py:13-55, test_categorical.py:24-161,
test_multinomial.py:11-31,104-117: the reference asserts equality with exactly these SciPy
calls), plus the CPU oracle's outputs on seeded larger cases (d = 64 config-4 block, batched
covariances, k = 1000 categories).  The reference itself cannot be imported here (PyTensor
absent).  Run from the repo root:  python tests/golden/make_golden_mv.py"""
import os
import sys

import numpy as np
from scipy import stats

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import restatement as R  # noqa: E402
from tests.test_oracle_pinning import MATN_CASES, MVN_CASES, MVT_NU  # noqa: E402

PROBS = [[0.5, 0.5], [0.25, 0.25, 0.25, 0.25], [0.1, 0.2, 0.3, 0.4], [0.7, 0.2, 0.1], [0.01, 0.01, 0.98]]
MULT = [(10, [0.5, 0.5]), (20, [0.2, 0.3, 0.5]), (15, [0.25] * 4), (30, [0.1, 0.2, 0.3, 0.4])]


def main():
    out = {}
    for i, (mu, cov) in enumerate(MVN_CASES):
        d = stats.multivariate_normal(mean=mu, cov=cov)
        xs = np.vstack([mu[None, :], d.rvs(size=10, random_state=814)])
        out[f"mvn{i}__mu"], out[f"mvn{i}__cov"], out[f"mvn{i}__x"] = mu, cov, xs
        out[f"mvn{i}__scipy_logpdf"] = d.logpdf(xs)
        out[f"mvn{i}__oracle_logpdf"] = R.mvnormal_logpdf(xs, mu, cov)
        t = stats.multivariate_t(df=MVT_NU[i], loc=mu, shape=cov)
        xt = np.vstack([mu[None, :], t.rvs(size=10, random_state=814)])
        out[f"mvt{i}__nu"], out[f"mvt{i}__x"] = np.float64(MVT_NU[i]), xt
        out[f"mvt{i}__scipy_logpdf"] = t.logpdf(xt)
        out[f"mvt{i}__oracle_logpdf"] = R.mvstudentt_logpdf(xt, MVT_NU[i], mu, cov)
    # config-4 shaped block: d = 64, 256 observations, seeds 6/7/8
    dd = 64
    A = np.random.default_rng(6).normal(size=(dd, dd))
    cov = A @ A.T / dd + np.eye(dd)
    mu = np.random.default_rng(7).normal(size=dd)
    X = mu + np.random.default_rng(8).normal(size=(256, dd)) @ np.linalg.cholesky(cov).T
    out["cfg4__mu"], out["cfg4__cov"], out["cfg4__x"] = mu, cov, X
    out["cfg4__scipy_logpdf"] = stats.multivariate_normal(mu, cov).logpdf(X)
    out["cfg4__oracle_logpdf"] = R.mvnormal_logpdf(X, mu, cov)
    out["cfg4__oracle_mvt_logpdf"] = R.mvstudentt_logpdf(X, 7.5, mu, cov)
    # batched covariances (mvnormal.py:40-41: cov may carry batch axes)
    rng = np.random.default_rng(20)
    B, d2 = 9, 12
    Ab = rng.normal(size=(B, d2, d2))
    covs = Ab @ np.swapaxes(Ab, -1, -2) / d2 + np.eye(d2)
    mus = rng.normal(size=(B, d2))
    Xb = rng.normal(size=(B, d2))
    out["bat__mu"], out["bat__cov"], out["bat__x"] = mus, covs, Xb
    out["bat__oracle_logpdf"] = R.mvnormal_logpdf(Xb, mus, covs)
    for i, (M, U, V) in enumerate(MATN_CASES):
        Xm = M + 0.1 * np.ones_like(M)
        out[f"matn{i}__M"], out[f"matn{i}__U"], out[f"matn{i}__V"], out[f"matn{i}__x"] = M, U, V, Xm
        out[f"matn{i}__scipy_logpdf"] = np.float64(stats.matrix_normal(mean=M, rowcov=U, colcov=V).logpdf(Xm))
        out[f"matn{i}__oracle_logpdf"] = R.matrixnormal_logpdf(Xm, M, U, V)
    for i, probs in enumerate(PROBS):
        k = len(probs)
        pr = np.asarray(probs) / np.sum(probs)
        d = stats.rv_discrete(values=(np.arange(k), pr))
        xs = np.array([-2, -1, 0, 1, k - 1, k, k + 1], dtype="float64")
        qs = np.array([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0])
        out[f"cat{i}__p"], out[f"cat{i}__x"], out[f"cat{i}__q"] = np.asarray(probs, float), xs, qs
        with np.errstate(all="ignore"):
            out[f"cat{i}__scipy_pdf"], out[f"cat{i}__scipy_logpdf"] = d.pmf(xs), d.logpmf(xs)
            out[f"cat{i}__scipy_cdf"], out[f"cat{i}__scipy_logcdf"] = d.cdf(xs), d.logcdf(xs)
            out[f"cat{i}__scipy_sf"], out[f"cat{i}__scipy_ppf"], out[f"cat{i}__scipy_isf"] = d.sf(xs), d.ppf(qs), d.isf(qs)
    rng = np.random.default_rng(30)
    p1000 = rng.dirichlet(np.ones(1000)) * 3.7
    xk = rng.integers(-5, 1005, 4096)
    qk = rng.uniform(0, 1, 4096)
    out["catk__p"], out["catk__x"], out["catk__q"] = p1000, xk, qk
    with np.errstate(all="ignore"):
        out["catk__oracle_logpdf"] = R.categorical_logpdf(xk.astype(float), p1000)
        out["catk__oracle_cdf"] = R.categorical_cdf(xk.astype(float), p1000)
        out["catk__oracle_ppf"] = R.categorical_ppf(qk, p1000)
    for i, (n, p) in enumerate(MULT):
        p = np.asarray(p)
        d = stats.multinomial(n=n, p=p)
        x = np.round(d.mean()).astype(int)
        x[-1] = n - x[:-1].sum()
        xs = np.vstack([x[None, :], d.rvs(size=40, random_state=5)])
        xs[-1, 0] += 1   # sum != n
        xs[-2, -1] = -1  # negative count
        out[f"mult{i}__n"], out[f"mult{i}__p"], out[f"mult{i}__x"] = np.int64(n), p, xs
        exp = d.logpmf(xs[:-2])
        out[f"mult{i}__scipy_logpdf"] = np.concatenate([exp, [-np.inf, -np.inf]])
        out[f"mult{i}__oracle_logpdf"] = R.multinomial_logpdf(xs, n, p)
    path = os.path.join(ROOT, "tests", "golden", "multivariate.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
