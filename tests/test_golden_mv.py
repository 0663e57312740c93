"""tests/golden/multivariate.npz (made by tests/golden/make_golden_mv.py): the CPU oracle
reproduces it (no GPU), and the HIP path reproduces it (`-m gpu`).  SciPy-derived expectations
use the reference's own tolerances (rtol 1e-5 multivariate, 1e-6 categorical, exact ppf);
oracle-derived ones the 1e-12 bar; support decisions (-inf) must match exactly."""

import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import restatement as R
from tests import harness as H

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multivariate.npz"))


class OracleNS:
    mvnormal_logpdf = staticmethod(R.mvnormal_logpdf)
    mvstudentt_logpdf = staticmethod(R.mvstudentt_logpdf)
    matrixnormal_logpdf = staticmethod(R.matrixnormal_logpdf)
    multinomial_logpdf = staticmethod(R.multinomial_logpdf)

    @staticmethod
    def categorical(fn, x, p):
        return R.TABLE[f"categorical_{fn}"](np.asarray(x, dtype=np.float64), p)


class HipNS:
    @staticmethod
    def mvnormal_logpdf(x, mu, cov):
        from distributions_amd import mvnormal
        return mvnormal.logpdf(x, mu, cov)

    @staticmethod
    def mvstudentt_logpdf(x, nu, mu, cov):
        from distributions_amd import mvstudentt
        return mvstudentt.logpdf(x, nu, mu, cov)

    @staticmethod
    def matrixnormal_logpdf(X, M, U, V):
        from distributions_amd import matrixnormal
        return matrixnormal.logpdf(X, M, U, V)

    @staticmethod
    def multinomial_logpdf(x, n, p):
        from distributions_amd import multinomial
        return multinomial.logpdf(x, n, p)

    @staticmethod
    def categorical(fn, x, p):
        from distributions_amd import categorical
        return getattr(categorical, fn)(x, p)


def check_all(ns):
    for i in range(4):
        mu, cov = G[f"mvn{i}__mu"], G[f"mvn{i}__cov"]
        got = ns.mvnormal_logpdf(G[f"mvn{i}__x"], mu, cov)
        assert_allclose(got, G[f"mvn{i}__scipy_logpdf"], rtol=1e-5)
        assert np.all(H.close(got, G[f"mvn{i}__oracle_logpdf"]))
        got = ns.mvstudentt_logpdf(G[f"mvt{i}__x"], float(G[f"mvt{i}__nu"]), mu, cov)
        assert_allclose(got, G[f"mvt{i}__scipy_logpdf"], rtol=1e-5)
        assert np.all(H.close(got, G[f"mvt{i}__oracle_logpdf"]))
    mu, cov, X = G["cfg4__mu"], G["cfg4__cov"], G["cfg4__x"]
    got = ns.mvnormal_logpdf(X, mu, cov)
    assert np.all(H.close(got, G["cfg4__scipy_logpdf"], 1e-10)) and np.all(H.close(got, G["cfg4__oracle_logpdf"]))
    assert np.all(H.close(ns.mvstudentt_logpdf(X, 7.5, mu, cov), G["cfg4__oracle_mvt_logpdf"]))
    assert np.all(H.close(ns.mvnormal_logpdf(G["bat__x"], G["bat__mu"], G["bat__cov"]), G["bat__oracle_logpdf"]))
    for i in range(3):
        got = ns.matrixnormal_logpdf(G[f"matn{i}__x"], G[f"matn{i}__M"], G[f"matn{i}__U"], G[f"matn{i}__V"])
        assert_allclose(got, G[f"matn{i}__scipy_logpdf"], rtol=1e-5)
        assert np.all(H.close(got, G[f"matn{i}__oracle_logpdf"]))
    with np.errstate(all="ignore"):
        for i in range(5):
            p, xs, qs = G[f"cat{i}__p"], G[f"cat{i}__x"], G[f"cat{i}__q"]
            for fn in ("pdf", "logpdf", "cdf", "logcdf"):
                assert_allclose(ns.categorical(fn, xs, p), G[f"cat{i}__scipy_{fn}"], rtol=1e-6, err_msg=fn)
            assert_allclose(ns.categorical("sf", xs, p), G[f"cat{i}__scipy_sf"], rtol=1e-6, atol=1e-15)
            assert np.array_equal(ns.categorical("ppf", qs, p), G[f"cat{i}__scipy_ppf"])
            assert np.array_equal(ns.categorical("isf", qs, p), G[f"cat{i}__scipy_isf"])
        p, xk, qk = G["catk__p"], G["catk__x"], G["catk__q"]
        got = ns.categorical("logpdf", xk, p)
        assert np.array_equal(np.isneginf(got), np.isneginf(G["catk__oracle_logpdf"]))
        assert np.all(H.close(got, G["catk__oracle_logpdf"]))
        assert np.all(H.close(ns.categorical("cdf", xk, p), G["catk__oracle_cdf"]))
        assert np.array_equal(ns.categorical("ppf", qk, p), G["catk__oracle_ppf"])
    for i in range(4):
        n, p, xs = int(G[f"mult{i}__n"]), G[f"mult{i}__p"], G[f"mult{i}__x"]
        got = ns.multinomial_logpdf(xs, n, p)
        assert np.array_equal(np.isneginf(got), np.isneginf(G[f"mult{i}__scipy_logpdf"]))
        assert_allclose(got, G[f"mult{i}__scipy_logpdf"], rtol=1e-5)
        assert np.all(H.close(got, G[f"mult{i}__oracle_logpdf"]))


def test_oracle_reproduces_golden_mv():
    check_all(OracleNS)


@pytest.mark.gpu
def test_hip_reproduces_golden_mv(gpu):
    check_all(HipNS)
