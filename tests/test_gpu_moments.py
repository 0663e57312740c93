"""GPU parity tests of the moment functions (m355_moment through distributions_amd.<dist>.mean / mode / median /
var / std / skewness / kurtosis / entropy) against oracle/restatement_moments.py, and the reference's own
acceptance bar (scipy.stats at the tolerances of tests/helper_scipy.py) on the HIP path."""

import importlib

import numpy as np
import pytest

from oracle import restatement_moments as M
from tests import harness as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def mod(name):
    return importlib.import_module(f"distributions_amd.{name}")


_CASES = H.grids() + H.moment_extra_cases()


@pytest.mark.parametrize("name, sp_dist, params, sp_params, support, kw", _CASES,
                         ids=[f"{g[0]}-{'-'.join(str(v) for v in g[2])}" for g in _CASES])
def test_moments_vs_oracle(gpu, name, sp_dist, params, sp_params, support, kw):
    """Every moment function at every parameter point of the reference's tests (+ branch points), scalar parameters:
    1e-12 for closed forms, the ppf tolerance for medians (tests/harness.py::moment_close)."""
    m = mod(name)
    for stat in M.STATS:
        exp = float(M.moment(name, stat, *params))
        got = getattr(m, stat)(*params)
        assert isinstance(got, np.ndarray) and got.shape == (), (name, stat, type(got))
        assert H.moment_close(name, stat, float(got), exp, params), (name, stat, params, float(got), exp)


def test_moments_golden_vectors(gpu):
    """The committed vectors (tests/golden/moments.json, made by tests/golden/make_golden_moments.py)."""
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "moments.json")) as f:
        cases = json.load(f)
    assert len(cases) > 250
    for c in cases:
        params = [float.fromhex(v) for v in c["params"]]
        m = mod(c["dist"])
        for stat, hx in c["values"].items():
            exp = float.fromhex(hx)
            got = float(getattr(m, stat)(*params))
            assert H.moment_close(c["dist"], stat, got, exp, params), (c["dist"], stat, params, got, exp)


def test_moments_reference_bar_vs_scipy(gpu):
    """tests/helper_scipy.py:45-48,163-238 of the reference on the HIP path: scipy.stats at rtol 1e-6 (entropy 1e-4)
    for the families with the default settings."""
    from numpy.testing import assert_allclose

    for name, sp_dist, params, sp_params, support, kw in H.grids():
        if name in ("rice", "vonmises", "negativebinomial", "exgaussian", "skewnormal", "skew_studentt", "truncatednormal", "halfstudentt",
                    "cauchy"):
            continue  # per-test tolerances / skips of the reference: covered on the oracle (tests/test_oracle_moments.py)
        d = sp_dist(**sp_params)
        m = mod(name)
        with np.errstate(all="ignore"):
            sk, ku = (float(v) for v in d.stats(moments="sk"))
            exp = {"mean": d.mean(), "median": d.median(), "var": d.var(), "std": d.std(), "skewness": sk, "kurtosis": ku,
                   "entropy": d.entropy()}
        for stat, e in exp.items():
            assert_allclose(getattr(m, stat)(*params), e, rtol=1e-4 if stat == "entropy" else 1e-6, atol=1e-15 if stat == "median" else 0,
                            err_msg=f"{name}.{stat} {params}")


def test_moments_batched_parameters(gpu):
    """Arrays of parameters: one value per parameter set of the broadcast shape, thread-per-set and wavefront-per-set
    kernels, float32 and torch inputs."""
    import torch

    from distributions_amd import binomial, gamma, rice, skew_studentt, zi_poisson

    rng = np.random.default_rng(5)
    a = rng.uniform(0.3, 40.0, (7, 1))
    b = rng.uniform(0.2, 5.0, (1, 33))
    for stat in M.STATS:
        got = getattr(gamma, stat)(a, b)
        exp = M.moment("gamma", stat, a, b)
        assert got.shape == (7, 33)
        tol = 1e-9 if stat == "median" else 1e-12
        assert np.all(np.abs(got - exp) <= tol * np.maximum(1.0, np.abs(exp))), stat
    # wavefront-per-parameter-set kernels: more parameter sets than wavefronts in a block, ragged sizes
    nu = rng.uniform(0.1, 6.0, 301)
    sg = rng.uniform(0.5, 2.0, 301)
    for stat in ("skewness", "kurtosis", "entropy"):
        got = getattr(rice, stat)(nu, sg)
        exp = M.moment("rice", stat, nu[:40], sg[:40])
        assert got.shape == (301,)
        assert np.all(np.abs(got[:40] - exp) <= 1e-9 * np.maximum(1.0, np.abs(exp))), stat
        one = getattr(rice, stat)(nu[300], sg[300])
        assert abs(float(one) - got[300]) <= 1e-13 * max(1.0, abs(got[300]))
    n = rng.integers(1, 400, 130).astype(np.float64)
    p = rng.uniform(0.05, 0.95, 130)
    got = binomial.entropy(n, p)
    exp = M.moment("binomial", "entropy", n, p)
    assert np.all(np.abs(got - exp) <= 1e-12 * np.maximum(1.0, np.abs(exp)))
    psi = rng.uniform(0.1, 1.0, 50)
    mu = rng.uniform(0.5, 30.0, 50)
    got = zi_poisson.entropy(psi, mu)
    exp = M.moment("zi_poisson", "entropy", psi, mu)
    assert np.all(np.abs(got - exp) <= 1e-12 * np.maximum(1.0, np.abs(exp)))
    # grid-search mode over 200 points and a 1000-point entropy per parameter set
    ab = rng.uniform(1.5, 9.0, (2, 64))
    got = skew_studentt.entropy(ab[0], ab[1], 0.5, 1.5)
    exp = M.moment("skew_studentt", "entropy", ab[0][:16], ab[1][:16], 0.5, 1.5)
    assert np.all(np.abs(got[:16] - exp) <= 1e-10 * np.maximum(1.0, np.abs(exp)))
    # float32 parameters give float32; torch in gives a device tensor out
    a32 = a.astype(np.float32)
    r32 = gamma.entropy(a32, np.float32(1.5))
    assert r32.dtype == np.float32 and np.allclose(r32, M.moment("gamma", "entropy", a32.astype(np.float64), 1.5), rtol=1e-5)
    t = gamma.mean(torch.tensor([2.0, 3.0], dtype=torch.float64, device="cuda"), 4.0)
    assert isinstance(t, torch.Tensor) and t.is_cuda and torch.allclose(t.cpu(), torch.tensor([0.5, 0.75], dtype=torch.float64))
    # empty batch
    assert gamma.var(np.zeros((0, 3)), 1.0).shape == (0, 3)
    # discreteweibull.mode is int64 (pt.cast, discreteweibull.py:19)
    from distributions_amd import discreteweibull

    assert discreteweibull.mode(np.array([0.5, 0.9]), np.array([1.3, 1.3])).dtype == np.int64


def test_moment_errors(gpu):
    from distributions_amd import _lib, gamma

    with pytest.raises(TypeError):
        gamma.mean(1.0)
    L = _lib.lib()
    assert L.m355_moment(_lib.DIST["gamma"], 99, _lib.F64, 1, 2, None, None, None) != 0
    assert b"moment" in L.m355_last_error()
