"""The PyTensor Op seam (distributions_amd/pytensor_ops.py).  PyTensor is absent in the authoring
container, so on such a box this file only checks that the optional module fails with a clear
ImportError and that nothing else in the package needs PyTensor; where PyTensor and a GPU exist
the Ops are evaluated against the oracle."""

import importlib

import numpy as np
import pytest

try:
    import pytensor  # noqa: F401
    HAVE = True
except ImportError:
    HAVE = False


def test_optional_dependency_boundary():
    import distributions_amd  # noqa: F401  (must import without PyTensor)

    if not HAVE:
        with pytest.raises(ImportError, match="PyTensor"):
            importlib.import_module("distributions_amd.pytensor_ops")
    from distributions_amd.csrc_support import SUPPORTED

    assert ("normal", "ppf") in SUPPORTED and ("poisson", "ppf") in SUPPORTED and len(SUPPORTED) == 8 * 46


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE, reason="PyTensor not installed")
def test_ops_against_oracle(gpu):
    import pytensor.tensor as pt

    from distributions_amd.pytensor_ops import symbolic
    from oracle import restatement as R
    from tests import harness as H

    x = np.random.default_rng(0).gamma(2.5, 1.0, (7, 33))
    xv = pt.dmatrix("x")
    g = symbolic("gamma")
    assert np.all(H.close(g.logpdf(xv, 2.5, 1.3).eval({xv: x}), R.gamma_logpdf(x, 2.5, 1.3)))
    assert np.all(H.close(g.ppf(pt.constant(np.array([0.1, 0.5])), 2.5, 1.3).eval(), R.gamma_ppf(np.array([0.1, 0.5]), 2.5, 1.3)))
    tot = g.sum_logpdf(xv, 2.5, 1.3).eval({xv: x})
    assert abs(tot - R.gamma_logpdf(x, 2.5, 1.3).sum()) <= 1e-12 * np.abs(R.gamma_logpdf(x, 2.5, 1.3)).sum()


@pytest.mark.gpu
@pytest.mark.skipif(not HAVE, reason="PyTensor not installed")
def test_op_gradients_against_oracle(gpu):
    """pytensor.grad through the custom Ops (L_op -> m355_logpdf_grad) against the gradient oracle."""
    import pytensor
    import pytensor.tensor as pt

    from distributions_amd.pytensor_ops import symbolic
    from oracle import restatement_grad as G

    x = np.random.default_rng(1).gamma(2.5, 1.0, (5, 9))
    xv, a, b = pt.dmatrix("x"), pt.dscalar("a"), pt.dvector("b")
    g = symbolic("gamma")
    bv = np.linspace(0.5, 2.0, 9)
    val, gr = G.logpdf_grad("gamma", x, 2.5, bv)
    for total in (g.sum_logpdf(xv, a, b), g.logpdf(xv, a, b).sum()):
        gx, ga, gb = pytensor.grad(total, [xv, a, b])
        ex, ea, eb = pytensor.function([xv, a, b], [gx, ga, gb])(x, 2.5, bv)
        assert np.allclose(ex, gr[0], rtol=1e-12, atol=1e-12)
        assert abs(ea - gr[1].sum()) <= 1e-11 * np.abs(gr[1]).sum()
        assert np.allclose(eb, gr[2].sum(axis=0), rtol=1e-11, atol=1e-11)


# ---- without PyTensor: the Ops run against tests/pytensor_standin.py ------------------------------------------------
@pytest.mark.skipif(HAVE, reason="real PyTensor present: the tests above cover the Ops")
def test_make_node_logic_with_standin():
    """dtype promotion, static broadcast shapes, arity and scalar-value errors of make_node (no GPU needed)."""
    from tests.pytensor_standin import installed

    with installed() as (ops, pytensor):
        pt = pytensor.tensor
        x, v32, k = pt.dmatrix("x"), pt.fmatrix("f"), pt.lvector("k")
        out = ops.M355Elemwise("gamma", "logpdf")(x, 2.5, pt.constant(np.ones((1, 7))))
        assert out.dtype == "float64" and out.type.shape == (None, 7) and out.owner.op.dist == "gamma"
        assert ops.M355Elemwise("normal", "logpdf")(v32, np.float32(0.0), np.float32(1.0)).dtype == "float32"
        assert ops.M355Elemwise("normal", "logpdf")(v32, 0.0, 1.0).dtype == "float64"  # a float64 constant promotes
        assert ops.M355Elemwise("poisson", "logpdf")(k, 3.0).dtype == "float64"          # integer observations
        assert ops.M355Elemwise("normal", "logpdf")(pt.constant(np.zeros((5, 1, 3))), pt.constant(np.zeros((4, 1))), 1.0).type.shape == (5, 4, 3)
        with pytest.raises(TypeError, match="takes 2 parameters"):
            ops.M355Elemwise("normal", "logpdf")(x, 0.0)
        with pytest.raises(ValueError, match="unknown"):
            ops.M355Elemwise("normal", "nosuchfn")
        with pytest.raises(NotImplementedError):
            ops.M355LogpdfPartials("bernoulli")
        with pytest.raises(ValueError, match="scalar"):
            ops.M355MvLogpdf(0)(pt.constant(1.0), pt.constant(np.zeros(2)), pt.constant(np.eye(2)))
        assert ops.M355MvLogpdf(0)(pt.dmatrix(), pt.dvector(), pt.dmatrix()).type.ndim == 1
        assert ops.M355SumLogp("gamma")(x, 2.5, 1.3).type.shape == ()
        ns = ops.symbolic("beta")
        assert {"logpdf", "logcdf", "logsf", "ppf", "pdf", "cdf", "sf", "isf", "sum_logpdf", "sum_logcdf", "sum_logsf",
                "rvs"} <= set(vars(ns))
        # rvs: a RandomVariable-shaped node -- outputs (next_rng, draw), static shape and dtype of the draw
        rng = pytensor.shared(np.random.default_rng(3))
        d = ops.symbolic("normal").rvs(0.0, pt.constant(np.ones((4, 1))), size=(5, 4, 3), random_state=rng)
        assert d.owner.outputs[1] is d and d.type.shape == (5, 4, 3) and d.dtype == "float64"
        assert type(d.owner.outputs[0].type) is type(rng.type) and d.owner.inputs[0] is rng
        assert ops.symbolic("poisson").rvs(3.0, size=7).dtype == "int64"
        assert ops.symbolic("gamma").rvs(pt.constant(np.ones((6, 1))), pt.constant(np.ones(2))).type.shape == (6, 2)
        assert ops.symbolic("multinomial").rvs(10, pt.constant(np.ones(4) / 4), size=3).type.shape == (3, 4)
        assert ops.symbolic("categorical").rvs(pt.constant(np.ones(4) / 4)).type.shape == ()
        with pytest.raises(TypeError, match="takes 2 parameters"):
            ops.symbolic("normal").rvs(0.0)
        with pytest.raises(TypeError, match="size"):
            ops.symbolic("normal").rvs(0.0, 1.0, size=pt.constant(np.ones((2, 2), dtype="int64")))
        assert ops.M355MatrixNormalLogpdf()(pt.dtensor3(), pt.dmatrix(), pt.dmatrix(), pt.dmatrix()).type.ndim == 1
        with pytest.raises(ValueError, match="rowcov"):
            ops.M355MatrixNormalLogpdf()(pt.dvector(), pt.dmatrix(), pt.dmatrix(), pt.dmatrix())
        assert ops.M355Categorical("logpdf")(pt.lvector(), pt.dvector()).type.shape == (None,)
        with pytest.raises(ValueError, match="probability vector"):
            ops.M355Categorical("logpdf")(pt.lvector(), pt.dmatrix())
        assert ops.M355MultinomialLogpdf()(pt.dmatrix(), 10, pt.dvector()).type.ndim == 1
    assert "pytensor" not in __import__("sys").modules or HAVE


@pytest.mark.gpu
@pytest.mark.skipif(HAVE, reason="real PyTensor present: the tests above cover the Ops")
def test_ops_perform_and_l_op_with_standin(gpu):
    """perform (C-ABI through _dispatch) and the L_op bodies -- M355LogpdfPartials / M355SumLogpGrad, the reduction
    of the per-element partials over the broadcast axes -- executed through the stand-in's eval / reverse mode and
    compared with the value and gradient oracles."""
    from oracle import restatement as R
    from oracle import restatement_grad as G
    from tests import harness as H
    from tests.pytensor_standin import installed

    rng = np.random.default_rng(0)
    x = rng.gamma(2.5, 1.0, (7, 33))
    with installed() as (ops, pytensor):
        pt = pytensor.tensor
        xv = pt.dmatrix("x")
        g = ops.symbolic("gamma")
        assert np.all(H.close(g.logpdf(xv, 2.5, 1.3).eval({xv: x}), R.gamma_logpdf(x, 2.5, 1.3)))
        q = np.array([0.1, 0.5, 0.9])
        assert np.all(H.close(g.ppf(pt.constant(q), 2.5, 1.3).eval(), R.gamma_ppf(q, 2.5, 1.3)))
        terms = R.gamma_logpdf(x, 2.5, 1.3)
        assert abs(g.sum_logpdf(xv, 2.5, 1.3).eval({xv: x}) - terms.sum()) <= 1e-12 * np.abs(terms).sum()
        x32 = x.astype(np.float32)
        r32 = ops.symbolic("normal").logpdf(pt.fmatrix("f"), np.float32(0.3), np.float32(1.7))
        got32 = r32.eval({r32.owner.inputs[0]: x32})
        assert got32.dtype == np.float32 and np.all(H.close(got32, R.normal_logpdf(x32.astype(np.float64), np.float32(0.3).item(), np.float32(1.7).item()), 1e-5))
        # multivariate
        mu, cov = np.array([0.5, -1.0]), np.array([[2.0, 0.3], [0.3, 1.0]])
        xm = rng.normal(size=(11, 2))
        mv = ops.symbolic("mvnormal")
        assert np.all(H.close(mv.logpdf(pt.constant(xm), pt.constant(mu), pt.constant(cov)).eval(), R.mvnormal_logpdf(xm, mu, cov)))
        # gradients: both routes (fused sum Op, and sum of the elementwise Op) against the gradient oracle
        x5 = rng.gamma(2.5, 1.0, (5, 9))
        bv = np.linspace(0.5, 2.0, 9)
        val, gr = G.logpdf_grad("gamma", x5, 2.5, bv)
        xs, a, b = pt.dmatrix("x"), pt.dscalar("a"), pt.dvector("b")
        b.type.shape = (9,)  # static shapes as PyTensor would have after shape inference
        xs.type.shape = (5, 9)
        for total in (g.sum_logpdf(xs, a, b), g.logpdf(xs, a, b).sum()):
            gx, ga, gb = pytensor.grad(total, [xs, a, b])
            ex, ea, eb = pytensor.function([xs, a, b], [gx, ga, gb])(x5, 2.5, bv)
            assert np.allclose(ex, gr[0], rtol=1e-12, atol=1e-12)
            assert abs(float(ea) - gr[1].sum()) <= 1e-11 * np.abs(gr[1]).sum()
            assert np.allclose(np.ravel(eb), gr[2].sum(axis=0), rtol=1e-11, atol=1e-11)
        with pytest.raises(NotImplementedError):
            pytensor.grad(g.logcdf(xs, a, b).sum(), [a])


@pytest.mark.gpu
@pytest.mark.skipif(HAVE, reason="real PyTensor present")
def test_rvs_and_multivariate_ops_with_standin(gpu):
    """The rvs Op threads the generator as pt.random does (normal.py:66-67; skewnormal.py:90-91 takes next_rng from
    `.owner.outputs` for its second draw); matrixnormal / categorical / multinomial Ops against the oracle."""
    from scipy import stats

    from oracle import restatement as R
    from tests import harness as H
    from tests.pytensor_standin import installed

    rng = np.random.default_rng(5)
    with installed() as (ops, pytensor):
        pt = pytensor.tensor
        normal = ops.symbolic("normal")
        src = pytensor.shared(np.random.default_rng(11))
        # the reference's idiom: two draws chained through next_rng are different streams; the same generator state
        # gives the same draw again; the input generator is not advanced (non-inplace Op)
        next_rng, u0 = normal.rvs(0.0, 1.0, size=(200_000,), random_state=src).owner.outputs
        v = normal.rvs(0.0, 1.0, size=(200_000,), random_state=next_rng)
        a, b = pytensor.function([], [u0, v])()
        assert a.shape == b.shape == (200_000,) and not np.array_equal(a, b)
        assert abs(np.corrcoef(a, b)[0, 1]) < 0.01
        assert stats.kstest(a, "norm").pvalue > 1e-4 and stats.kstest(b, "norm").pvalue > 1e-4
        assert np.array_equal(u0.eval(), a)  # same generator state in, same draw out
        assert src.value.bit_generator.state == np.random.default_rng(11).bit_generator.state
        # parameters broadcast against size; discrete families give int64
        g = ops.symbolic("gamma").rvs(pt.constant(np.array([[2.0], [8.0]])), 2.0, size=(2, 50_000), random_state=src).eval()
        assert g.shape == (2, 50_000) and abs(g[0].mean() - 1.0) < 0.02 and abs(g[1].mean() - 4.0) < 0.04
        k = ops.symbolic("poisson").rvs(4.0, size=100_000, random_state=src).eval()
        assert k.dtype == np.int64 and abs(k.mean() - 4.0) < 0.03
        p = np.array([0.1, 0.2, 0.3, 0.4])
        c = ops.symbolic("categorical").rvs(pt.constant(p), size=100_000, random_state=src).eval()
        assert c.dtype == np.int64 and np.allclose(np.bincount(c, minlength=4) / c.size, p, atol=0.01)
        m = ops.symbolic("multinomial").rvs(20, pt.constant(p), size=(1000,), random_state=src).eval()
        assert m.shape == (1000, 4) and np.all(m.sum(axis=1) == 20)
        # densities
        x = rng.integers(-1, 6, 300)
        cat = ops.symbolic("categorical")
        for fn in ("logpdf", "cdf", "logsf"):
            with np.errstate(all="ignore"):
                assert np.all(H.close(getattr(cat, fn)(pt.constant(x), pt.constant(p)).eval(), getattr(R, f"categorical_{fn}")(x, p))), fn
        counts = rng.multinomial(20, p, size=64)
        mn = ops.symbolic("multinomial")
        assert np.all(H.close(mn.logpdf(pt.constant(counts), 20, pt.constant(p)).eval(), R.multinomial_logpdf(counts, 20, p)))
        A = rng.normal(size=(3, 3))
        B = rng.normal(size=(4, 4))
        rowcov, colcov = A @ A.T + 3 * np.eye(3), B @ B.T + 4 * np.eye(4)
        X, mu = rng.normal(size=(9, 3, 4)), rng.normal(size=(3, 4))
        got = ops.symbolic("matrixnormal").logpdf(pt.constant(X), pt.constant(mu), pt.constant(rowcov), pt.constant(colcov)).eval()
        assert np.all(H.close(got, R.matrixnormal_logpdf(X, mu, rowcov, colcov)))
        xs = pt.dvector("x")
        tot = ops.symbolic("gamma").sum_logcdf(xs, 2.5, 1.3).eval({xs: g[0]})
        terms = R.gamma_logcdf(g[0], 2.5, 1.3)
        assert abs(tot - terms.sum()) <= 1e-12 * np.abs(terms).sum()
