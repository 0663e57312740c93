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
        assert {"logpdf", "logcdf", "logsf", "ppf", "pdf", "cdf", "sf", "isf", "sum_logpdf"} <= set(vars(ns))
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
