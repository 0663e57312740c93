"""Pins the gradient oracle (oracle/restatement_grad.py).  The reference holds no
gradient test at all (its gradients come from pytensor.grad of the logpdf graphs), so the pin is
(1) the value component of the dual-number restatement equals the pinned logpdf oracle, and
(2) every partial derivative equals a 5-point central difference of that pinned oracle wherever the
difference quotient is itself reliable (two step sizes agree).  CPU only."""

import numpy as np
import pytest

from oracle import restatement as R
from oracle import restatement_grad as G
from tests import harness as H

CASES = [g for g in H.grids() if g[0] in G.GRAD_TABLE]


def five_point(f, args, i, h):
    def at(t):
        a = [v.copy() for v in args]
        a[i] = a[i] + t * h
        with np.errstate(all="ignore"):
            return f(*a)

    return (at(-2.0) - 8.0 * at(-1.0) + 8.0 * at(1.0) - at(2.0)) / (12.0 * h)


@pytest.mark.parametrize("name, sp_dist, params, sp_params, support, kw", CASES,
                         ids=[f"{g[0]}-{'-'.join(str(v) for v in g[2])}" for g in CASES])
def test_grad_oracle_value_and_finite_differences(name, sp_dist, params, sp_params, support, kw):
    rng = np.random.default_rng(23)
    d = sp_dist(**sp_params)
    xs = np.asarray(d.rvs(400, random_state=rng), dtype=np.float64)
    val, grads = G.logpdf_grad(name, xs, *params)
    f = R.TABLE[f"{name}_logpdf"]
    with np.errstate(all="ignore"):
        ref = f(xs, *params)
    assert np.all(H.close(val, ref, 1e-13))
    assert len(grads) == 1 + len(params)
    args = [xs] + [np.full_like(xs, p) for p in params]
    integer_args = {"poisson": (0,), "binomial": (0, 1)}.get(name, ())
    checked = 0
    for i in range(len(args)):
        if i in integer_args and name == "binomial" and i == 1:
            continue  # n: the support switch x > n makes the quotient one-sided at x = n; checked through x below
        spread = float(np.std(xs)) or 1.0  # keeps the step inside the distribution's own scale (sigma = 1e-6 grids)
        h1 = 1e-3 * np.minimum(np.maximum(np.abs(args[i]), 1e-3), spread)
        fd1, fd2 = five_point(f, args, i, h1), five_point(f, args, i, 0.5 * h1)
        with np.errstate(all="ignore"):
            reliable = np.isfinite(fd1) & np.isfinite(fd2) & (np.abs(fd1 - fd2) <= 1e-8 * np.maximum(1.0, np.abs(fd2)))
            ok = np.abs(grads[i] - fd2) <= 1e-6 * np.maximum(1.0, np.abs(fd2))
        assert np.all(ok[reliable]), (name, params, i, xs[reliable & ~ok][:3], grads[i][reliable & ~ok][:3],
                                      fd2[reliable & ~ok][:3])
        checked += int(reliable.sum())
    assert checked > 0


def test_grad_outside_support_is_zero():
    """The -inf arm of the support switch is a constant: every partial is 0 there."""
    for name, params, x in (("gamma", (2.5, 1.3), -1.0), ("beta", (2.0, 5.0), 1.5), ("exponential", (0.3,), -0.1),
                            ("pareto", (2.5, 2.0), 1.0), ("uniform", (-1.0, 1.0), 3.0), ("weibull", (2.0, 5.0), -2.0)):
        v, g = G.logpdf_grad(name, np.array([x]), *params)
        assert np.isneginf(v[0]) and all(gi[0] == 0.0 for gi in g), name


def test_grad_structural_paths():
    """A partial only travels along graph paths that contain its variable (reverse-mode behaviour): at x = 0
    gamma's d/dalpha is log(0) = -inf, not NaN from (alpha - 1)/0 * 0."""
    v, g = G.logpdf_grad("gamma", np.array([0.0]), 2.5, 1.3)
    assert np.isneginf(g[1][0]) and g[2][0] == 2.5 / 1.3
    v, g = G.logpdf_grad("uniform", np.array([0.25]), 0.0, 2.0)
    assert g[0][0] == 0.0 and g[1][0] == 0.5 and g[2][0] == -0.5
