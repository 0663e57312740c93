"""GPU parity tests of the gradient path (m355_logpdf_grad through distributions_amd.<dist>.sum_logpdf_grad /
dlogpdf) against oracle/restatement_grad.py.  Tolerance 1e-12 in the sense of
tests/harness.py::grad_close (fp64), 1e-5 for fp32 data.  The reference pins no gradient: the oracle is pinned
by tests/test_oracle_grad.py (value = pinned logpdf oracle, partials = finite differences of it)."""

import math

import numpy as np
import pytest

from distributions_amd._lib import GRAD
from oracle import restatement_grad as G
from tests import harness as H

pytestmark = pytest.mark.gpu

CASES = [g for g in H.grids() if g[0] in GRAD]


def mod(name):
    import importlib

    return importlib.import_module(f"distributions_amd.{name}")


@pytest.mark.parametrize("name, sp_dist, params, sp_params, support, kw", CASES,
                         ids=[f"{g[0]}-{'-'.join(str(v) for v in g[2])}" for g in CASES])
def test_grad_dense_parity(gpu, name, sp_dist, params, sp_params, support, kw):
    rng = np.random.default_rng(31)
    d = sp_dist(**sp_params)
    xs = np.concatenate([np.asarray(d.rvs(20_000, random_state=rng), dtype=np.float64), H.extended_vals(d, support)])
    xs = xs[np.isfinite(xs)]
    val, grads = G.logpdf_grad(name, xs, *params)
    m = mod(name)
    got = m.dlogpdf(xs, *params)
    assert len(got) == 1 + len(params) and all(g.shape == xs.shape and g.dtype == np.float64 for g in got)
    assert H.grad_close([val] + list(got), [val] + grads, (name, params, xs))
    # fused sums over the in-support points (scalar parameters: the vectorised kernel when aligned)
    fin = np.isfinite(val) & np.all(np.isfinite(np.stack(grads)), axis=0)
    xi = np.ascontiguousarray(xs[fin])
    total, g = m.sum_logpdf_grad(xi, *params)
    assert abs(total - math.fsum(val[fin])) <= 1e-12 * max(1.0, float(np.abs(val[fin]).sum()))
    assert g[0].shape == xi.shape
    scale = np.maximum(1.0, np.max(np.abs(np.stack(grads)[:, fin]), axis=0))
    assert np.all(np.abs(g[0] - grads[0][fin]) <= 1e-12 * scale)
    for j in range(len(params)):
        e = grads[1 + j][fin]
        assert np.shape(g[1 + j]) == ()
        assert abs(float(g[1 + j]) - math.fsum(e)) <= 1e-12 * max(1.0, float(scale.sum())), (name, j)
    t2, g2 = m.sum_logpdf_grad(xi, *params)
    assert t2 == total and all(np.array_equal(a, b) for a, b in zip(g, g2))  # deterministic
    # x as data only (the x-partials are compiled out): same value and parameter gradients
    t3, g3 = m.sum_logpdf_grad(xi, *params, x_grad=False)
    assert g3[0] is None and abs(t3 - total) <= 1e-12 * max(1.0, float(np.abs(val[fin]).sum()))
    for j in range(len(params)):
        assert abs(float(g3[1 + j]) - float(g[1 + j])) <= 1e-12 * max(1.0, float(scale.sum())), (name, j)


def test_grad_layouts(gpu):
    """Array parameters: gradients come back in the shape of their argument, summed over the broadcast axes."""
    from distributions_amd import gamma, normal, studentt

    rng = np.random.default_rng(32)
    x = rng.gamma(2.5, 1.0, (300, 64))
    a_row = rng.uniform(0.5, 9.0, 64)
    b_col = rng.uniform(0.3, 3.0, (300, 1))
    val, gr = G.logpdf_grad("gamma", x, a_row, b_col)
    total, (gx, ga, gb) = gamma.sum_logpdf_grad(x, a_row, b_col)
    assert gx.shape == x.shape and ga.shape == a_row.shape and gb.shape == b_col.shape
    assert abs(total - math.fsum(val.ravel())) <= 1e-12 * float(np.abs(val).sum())
    assert np.allclose(gx, gr[0], rtol=1e-12, atol=1e-12)
    assert np.allclose(ga, gr[1].sum(axis=0), rtol=1e-11, atol=1e-10)
    assert np.allclose(gb, gr[2].sum(axis=1, keepdims=True), rtol=1e-11, atol=1e-10)
    # scalar x against a parameter vector; 0-d everything
    mu = rng.normal(0, 1, 50)
    val, gr = G.logpdf_grad("normal", 0.25, mu, 1.5)
    total, (gx, gm, gs) = normal.sum_logpdf_grad(0.25, mu, 1.5)
    assert np.shape(gx) == () and gm.shape == (50,) and np.shape(gs) == ()
    assert np.allclose(gm, gr[1], rtol=1e-12, atol=1e-12)
    assert abs(float(gx) - gr[0].sum()) <= 1e-11 and abs(float(gs) - gr[2].sum()) <= 1e-11
    total, g = normal.sum_logpdf_grad(0.25, 0.0, 1.0)
    v0, g0 = G.logpdf_grad("normal", 0.25, 0.0, 1.0)
    assert abs(total - float(v0)) <= 1e-12 and all(abs(float(a) - float(b)) <= 1e-12 for a, b in zip(g, g0))
    # unaligned view -> generic kernel, same numbers as the vectorised one
    xs = rng.standard_t(5, 40_001) * 2.0
    t1, g1 = studentt.sum_logpdf_grad(xs[1:], 5.0, 0.0, 2.0)
    t2, g2 = studentt.sum_logpdf_grad(np.ascontiguousarray(xs[1:]), 5.0, 0.0, 2.0)
    assert abs(t1 - t2) <= 1e-12 * abs(t2)
    # empty input
    total, g = normal.sum_logpdf_grad(np.zeros((0, 3)), 0.0, 1.0)
    assert total == 0.0 and g[0].shape == (0, 3) and float(g[1]) == 0.0


def test_grad_dtypes_and_torch(gpu):
    import torch

    from distributions_amd import gamma, poisson

    rng = np.random.default_rng(33)
    x32 = rng.gamma(2.5, 1.0, 10_001).astype(np.float32)
    val, gr = G.logpdf_grad("gamma", x32.astype(np.float64), 2.5, np.float64(np.float32(1.3)))
    total, g = gamma.sum_logpdf_grad(x32, np.float32(2.5), np.float32(1.3))
    assert g[0].dtype == np.float32
    assert np.all(np.abs(g[0] - gr[0]) <= 1e-5 * np.maximum(1.0, np.abs(gr[0])))
    assert abs(total - val.sum()) <= 1e-9 * np.abs(val).sum()
    xt = torch.from_numpy(rng.gamma(2.5, 1.0, 4096)).cuda()
    total, g = gamma.sum_logpdf_grad(xt, 2.5, 1.3)
    assert isinstance(total, torch.Tensor) and total.is_cuda and g[0].is_cuda
    val, gr = G.logpdf_grad("gamma", xt.cpu().numpy(), 2.5, 1.3)
    assert abs(float(total) - math.fsum(val)) <= 1e-12 * float(np.abs(val).sum())
    assert abs(float(g[1]) - math.fsum(gr[1])) <= 1e-12 * float(np.abs(gr[1]).sum())
    # integer observations: gradient w.r.t. the parameter is what a sampler uses
    k = rng.poisson(7.5, 5000)
    val, gr = G.logpdf_grad("poisson", k.astype(np.float64), 7.5)
    total, g = poisson.sum_logpdf_grad(k, 7.5)
    assert abs(float(g[1]) - math.fsum(gr[1])) <= 1e-12 * float(np.abs(gr[1]).sum())
    with pytest.raises(NotImplementedError):
        from distributions_amd import _dispatch

        _dispatch.logpdf_grad("bernoulli", k, 0.5)


def test_grad_full_size_1e8(gpu):
    """N = 1e8 fp64: the sums of the fused kernel against torch reductions of the elementwise partials."""
    import torch

    from distributions_amd import gamma

    n = 100_000_000
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = -torch.log1p(-torch.rand(n, dtype=torch.float64, device="cuda", generator=gen)) * 2.0
    total, (gx, ga, gb) = gamma.sum_logpdf_grad(x, 2.5, 1.3)
    xs, gxs = x[::97].cpu().numpy(), gx[::97].cpu().numpy()
    val, gr = G.logpdf_grad("gamma", xs, 2.5, 1.3)
    assert np.all(np.abs(gxs - gr[0]) <= 1e-12 * np.maximum(1.0, np.abs(gr[0])))
    ref_total = gamma.sum_logpdf(x, 2.5, 1.3)
    assert abs(float(total) - float(ref_total)) <= 1e-12 * n
    # d/dbeta = alpha/beta - x, d/dalpha = log x + log beta - digamma(alpha): closed forms summed by torch
    exp_b = n * 2.5 / 1.3 - float(x.sum())
    assert abs(float(gb) - exp_b) <= 1e-12 * float(x.sum())
    from scipy.special import digamma

    exp_a = float(torch.log(x).sum()) + n * (math.log(1.3) - digamma(2.5))
    assert abs(float(ga) - exp_a) <= 1e-11 * n
