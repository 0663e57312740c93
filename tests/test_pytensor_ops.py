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
