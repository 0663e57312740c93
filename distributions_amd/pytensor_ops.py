"""PyTensor `Op`s over the C-ABI: the seam a PyTensor graph uses to reach the HIP kernels
(BASELINE.json: "Python host code lowers each call to a custom PyTensor Op").

PyTensor is an optional dependency of this module only: importing it without PyTensor raises
ImportError with a clear message, nothing else in the package depends on it.  PyTensor is not
installable in the authoring environment, so these Ops are written against the
documented `Op` protocol (make_node / perform / infer_shape) and are exercised only where
PyTensor exists (tests/test_pytensor_ops.py skips otherwise).  `perform` receives host NumPy
arrays: this route pays PCIe; callers that keep data on the GPU use the eager functions with torch
tensors instead.

    from distributions_amd.pytensor_ops import symbolic
    normal = symbolic("normal")            # same function names as pytensor_distributions.normal
    lp = normal.logpdf(x_var, mu_var, sigma_var)          # a TensorVariable; .eval() runs on the GPU
"""

from __future__ import annotations

import types

import numpy as np

try:
    import pytensor
    import pytensor.tensor as pt
    from pytensor.graph.basic import Apply
    from pytensor.graph.op import Op
except ImportError as e:  # pragma: no cover - PyTensor absent here
    raise ImportError("distributions_amd.pytensor_ops needs PyTensor (pip install pytensor); the eager "
                      "functions in distributions_amd.<dist> do not") from e

from . import _catmult, _dispatch, _mv
from ._lib import DISCRETE, FN, GRAD, NPARAMS


def _out_dtype(inputs) -> str:
    """The reference's promotion: float32 only if every floating input is float32."""
    fl = [i.dtype for i in inputs if str(i.dtype).startswith("float")]
    return "float32" if fl and all(d == "float32" for d in fl) else "float64"


def _broadcast_static_shape(inputs):
    nd = max(i.type.ndim for i in inputs)
    shape = []
    for ax in range(nd):
        dims = []
        for i in inputs:
            k = ax - (nd - i.type.ndim)
            if k >= 0:
                dims.append(i.type.shape[k])
        known = [d for d in dims if d is not None and d != 1]
        shape.append(known[0] if known else (1 if all(d == 1 for d in dims) else None))
    return tuple(shape)


class M355Elemwise(Op):
    """out = <dist>.<fn>(x, *params), broadcast NumPy-style (m355_elemwise)."""

    __props__ = ("dist", "fn")

    def __init__(self, dist: str, fn: str):
        if dist not in NPARAMS or fn not in FN:
            raise ValueError(f"unknown (dist, fn) = ({dist}, {fn})")
        self.dist, self.fn = dist, fn

    def make_node(self, x, *params):
        if len(params) != NPARAMS[self.dist]:
            raise TypeError(f"{self.dist}.{self.fn} takes {NPARAMS[self.dist]} parameters, got {len(params)}")
        ins = [pt.as_tensor_variable(v) for v in (x, *params)]
        out = pt.tensor(dtype=_out_dtype(ins), shape=_broadcast_static_shape(ins))
        return Apply(self, ins, [out])

    def perform(self, node, inputs, output_storage):
        res = _dispatch.elemwise(self.dist, self.fn, *inputs)
        output_storage[0][0] = np.asarray(res, dtype=node.outputs[0].dtype)

    def infer_shape(self, fgraph, node, shapes):
        return [pt.broadcast_shape(*shapes, arrays_are_shapes=True)]

    def L_op(self, inputs, outputs, output_grads):
        """logpdf only: g_arg = sum over broadcast axes of (output gradient * d logpdf / d arg)."""
        if self.fn != "logpdf" or self.dist not in GRAD:
            raise NotImplementedError(f"no gradient kernel for {self.dist}.{self.fn}")
        partials = M355LogpdfPartials(self.dist)(*inputs)
        return [_sum_to(output_grads[0] * d, i) for d, i in zip(partials, inputs)]


class M355LogpdfPartials(Op):
    """[d logpdf/dx, d logpdf/dparam_0, ...] per element of the broadcast shape (m355_logpdf_grad): the
    factors the L_op of the elementwise logpdf multiplies with the output gradient."""

    __props__ = ("dist",)

    def __init__(self, dist: str):
        if dist not in GRAD:
            raise NotImplementedError(f"{dist} has no gradient kernel")
        self.dist = dist

    def make_node(self, x, *params):
        ins = [pt.as_tensor_variable(v) for v in (x, *params)]
        shape = _broadcast_static_shape(ins)
        dt = _out_dtype(ins)
        return Apply(self, ins, [pt.tensor(dtype=dt, shape=shape) for _ in ins])

    def perform(self, node, inputs, output_storage):
        _total, grads = _dispatch.logpdf_grad(self.dist, *inputs, elementwise=True)
        for st, g, out in zip(output_storage, grads, node.outputs):
            st[0] = np.asarray(g, dtype=out.dtype)

    def infer_shape(self, fgraph, node, shapes):
        return [pt.broadcast_shape(*shapes, arrays_are_shapes=True)] * len(shapes)


class M355SumLogpGrad(Op):
    """Gradients of sum(logpdf) w.r.t. (x, *params), each in the shape of its argument, from one fused pass."""

    __props__ = ("dist",)

    def __init__(self, dist: str):
        if dist not in GRAD:
            raise NotImplementedError(f"{dist} has no gradient kernel")
        self.dist = dist

    def make_node(self, x, *params):
        ins = [pt.as_tensor_variable(v) for v in (x, *params)]
        outs = [pt.tensor(dtype=_out_dtype(ins), shape=i.type.shape) for i in ins]
        return Apply(self, ins, outs)

    def perform(self, node, inputs, output_storage):
        _total, grads = _dispatch.logpdf_grad(self.dist, *inputs)
        for st, g, out in zip(output_storage, grads, node.outputs):
            st[0] = np.asarray(g, dtype=out.dtype)

    def infer_shape(self, fgraph, node, shapes):
        return list(shapes)


def _sum_to(g, like):
    """Sum `g` (broadcast shape) down to the shape of the variable `like` (Elemwise L_op convention)."""
    lead = g.type.ndim - like.type.ndim
    if lead:
        g = g.sum(axis=tuple(range(lead)))
    axes = [a for a in range(like.type.ndim) if like.type.shape[a] == 1 and g.type.shape[a] != 1]
    if axes:
        g = g.sum(axis=axes, keepdims=True)
    return g


class M355SumLogp(Op):
    """scalar = sum(<dist>.<fn>(x, *params)) in one fused pass (m355_sumlogp)."""

    __props__ = ("dist", "fn")

    def __init__(self, dist: str, fn: str = "logpdf"):
        self.dist, self.fn = dist, fn

    def make_node(self, x, *params):
        ins = [pt.as_tensor_variable(v) for v in (x, *params)]
        return Apply(self, ins, [pt.tensor(dtype="float64", shape=())])

    def perform(self, node, inputs, output_storage):
        output_storage[0][0] = np.asarray(_dispatch.sumlogp(self.dist, self.fn, *inputs), dtype="float64")

    def infer_shape(self, fgraph, node, shapes):
        return [()]

    def L_op(self, inputs, outputs, output_grads):
        if self.fn != "logpdf" or self.dist not in GRAD:
            raise NotImplementedError(f"no gradient kernel for sum of {self.dist}.{self.fn}")
        grads = M355SumLogpGrad(self.dist)(*inputs)
        return [output_grads[0] * g for g in grads]


class M355MvLogpdf(Op):
    """mvnormal.logpdf(x, mu, cov) (kind 0) / mvstudentt.logpdf(x, nu, mu, cov) (kind 1)."""

    __props__ = ("kind",)

    def __init__(self, kind: int):
        self.kind = kind

    def make_node(self, x, mu, cov, nu=0.0):
        ins = [pt.as_tensor_variable(v) for v in (x, mu, cov, nu)]
        if ins[0].type.ndim == 0:
            raise ValueError("Value can't be a scalar")  # mvnormal.py:19-20
        nd = max(ins[0].type.ndim - 1, ins[1].type.ndim - 1, ins[2].type.ndim - 2)
        if ins[0].type.ndim == 1 and nd == 0:
            nd = 0
        return Apply(self, ins, [pt.tensor(dtype="float64", shape=(None,) * nd)])

    def perform(self, node, inputs, output_storage):
        x, mu, cov, nu = inputs
        output_storage[0][0] = np.asarray(_mv.mv_logpdf(self.kind, x, mu, cov, float(nu)), dtype="float64")


class M355MatrixNormalLogpdf(Op):
    """matrixnormal.logpdf(X, mu, rowcov, colcov) (matrixnormal.py:68-94): X (..., n, p), one value per matrix."""

    __props__ = ()

    def make_node(self, X, mu, rowcov, colcov):
        ins = [pt.as_tensor_variable(v) for v in (X, mu, rowcov, colcov)]
        if ins[0].type.ndim < 2 or ins[2].type.ndim != 2 or ins[3].type.ndim != 2:
            raise ValueError("matrixnormal: X (..., n, p), rowcov (n, n), colcov (p, p)")
        nd = max(ins[0].type.ndim, ins[1].type.ndim) - 2
        return Apply(self, ins, [pt.tensor(dtype="float64", shape=(None,) * nd)])

    def perform(self, node, inputs, output_storage):
        output_storage[0][0] = np.asarray(_mv.matrixnormal_logpdf(*inputs), dtype="float64")


class M355Categorical(Op):
    """categorical.<fn>(x, p) for one probability vector p (categorical.py:60-112): the shape of x, float64."""

    __props__ = ("fn",)

    def __init__(self, fn: str):
        if fn not in FN:
            raise ValueError(f"unknown function {fn}")
        self.fn = fn

    def make_node(self, x, p):
        x, p = pt.as_tensor_variable(x), pt.as_tensor_variable(p)
        if p.type.ndim != 1:
            raise ValueError("categorical: p must be one probability vector of shape (k,)")
        return Apply(self, [x, p], [pt.tensor(dtype="float64", shape=x.type.shape)])

    def perform(self, node, inputs, output_storage):
        output_storage[0][0] = np.asarray(_catmult.categorical(self.fn, *inputs), dtype="float64")

    def infer_shape(self, fgraph, node, shapes):
        return [shapes[0]]


class M355MultinomialLogpdf(Op):
    """multinomial.logpdf(x, n, p) (multinomial.py:39-54): x (..., k), p (..., k), one value per count vector."""

    __props__ = ()

    def make_node(self, x, n, p):
        ins = [pt.as_tensor_variable(v) for v in (x, n, p)]
        if ins[0].type.ndim < 1 or ins[2].type.ndim < 1:
            raise ValueError("multinomial: x (..., k) and p (..., k) must share the last axis")
        nd = max(ins[0].type.ndim - 1, ins[1].type.ndim, ins[2].type.ndim - 1)
        return Apply(self, ins, [pt.tensor(dtype="float64", shape=(None,) * nd)])

    def perform(self, node, inputs, output_storage):
        output_storage[0][0] = np.asarray(_catmult.multinomial_logpdf(*inputs), dtype="float64")


class M355Rvs(Op):
    """(next_rng, draw) = <dist>.rvs(*params, size=size, random_state=rng): the two outputs of a PyTensor RandomVariable
    node, so that the reference's threading idiom works unchanged (normal.py:66-67 passes rng=random_state;
    skewnormal.py:90-91 takes `next_rng, u_0 = pt.random.normal(...).owner.outputs` and feeds next_rng to the next draw).
    inputs: rng (RandomGeneratorType; a fresh shared generator when random_state is None), size (int64 vector; empty with
    size_none), the parameters.  perform copies the generator (PyTensor's non-inplace convention), takes one 63-bit
    Philox seed from the copy -- which advances it: the next node draws another stream -- and returns (copy, draw).
    Draws are Philox counter-based on the GPU: distribution-equal to the reference's NumPy streams, not stream-equal
    (SURVEY.md quirk Q3)."""

    __props__ = ("dist", "size_none")

    def __init__(self, dist: str, size_none: bool):
        if dist not in NPARAMS and dist not in ("categorical", "multinomial"):
            raise ValueError(f"unknown distribution {dist}")
        self.dist, self.size_none = dist, bool(size_none)

    def make_node(self, rng, size, *params):
        nparams = {"categorical": 1, "multinomial": 2}.get(self.dist) or NPARAMS[self.dist]
        if len(params) != nparams:
            raise TypeError(f"{self.dist}.rvs takes {nparams} parameters, got {len(params)}")
        if rng is None:
            rng = pytensor.shared(np.random.default_rng())
        ps = [pt.as_tensor_variable(p) for p in params]
        size = pt.as_tensor_variable(np.zeros(0, dtype="int64") if self.size_none else np.atleast_1d(np.asarray(size)).astype("int64")
                                     if not hasattr(size, "type") else size)
        if size.type.ndim != 1 or not str(size.dtype).startswith("int"):
            raise TypeError("size must be None, an int, or a vector of ints")
        tail = (ps[-1].type.shape[-1],) if self.dist == "multinomial" else ()
        if self.size_none:
            shape = () if self.dist in ("categorical", "multinomial") else _broadcast_static_shape(ps)
        else:
            const = getattr(size, "data", getattr(size, "value", None))
            if const is None and size.type.shape[0] is None:
                raise TypeError("size must have a static length")
            shape = tuple(int(v) for v in const) if const is not None else (None,) * size.type.shape[0]
        dtype = "int64" if self.dist in DISCRETE or self.dist in ("categorical", "multinomial") else "float64"
        return Apply(self, [rng, size, *ps], [rng.type(), pt.tensor(dtype=dtype, shape=tuple(shape) + tail)])

    def perform(self, node, inputs, output_storage):
        import copy

        rng, size, *params = inputs
        rng = copy.deepcopy(rng)
        size = None if self.size_none else tuple(int(v) for v in size)
        if self.dist == "categorical":
            draw = _catmult.categorical_rvs(*params, size=size, random_state=rng)
        elif self.dist == "multinomial":
            draw = _catmult.multinomial_rvs(*params, size=size, random_state=rng)
        else:
            draw = _dispatch.rvs(self.dist, params, size=size, random_state=rng)
        output_storage[0][0] = rng
        output_storage[1][0] = np.asarray(draw, dtype=node.outputs[1].dtype)


def _rvs(dist):
    def rvs(*params, size=None, random_state=None):
        """The draw variable; `.owner.outputs` is (next_rng, draw) as with pt.random.*."""
        return M355Rvs(dist, size is None)(random_state, size, *params)[1]

    return rvs


def symbolic(dist: str) -> types.SimpleNamespace:
    """Namespace with the reference's function names for one distribution, each building an Op."""
    ns = types.SimpleNamespace()
    if dist == "matrixnormal":
        ns.logpdf = lambda X, mu, rowcov, colcov: M355MatrixNormalLogpdf()(X, mu, rowcov, colcov)
        ns.pdf = lambda X, mu, rowcov, colcov: pt.exp(ns.logpdf(X, mu, rowcov, colcov))
        return ns
    if dist == "categorical":
        for fn in FN:
            setattr(ns, fn, (lambda f: lambda x, p: M355Categorical(f)(x, p))(fn))
        ns.rvs = _rvs(dist)
        return ns
    if dist == "multinomial":
        ns.logpdf = lambda x, n, p: M355MultinomialLogpdf()(x, n, p)
        ns.pdf = lambda x, n, p: pt.exp(ns.logpdf(x, n, p))
        ns.rvs = _rvs(dist)
        return ns
    if dist == "mvnormal":
        ns.logpdf = lambda x, mu, cov: M355MvLogpdf(0)(x, mu, cov)
        ns.pdf = lambda x, mu, cov: pt.exp(ns.logpdf(x, mu, cov))
        return ns
    if dist == "mvstudentt":
        ns.logpdf = lambda x, nu, mu, cov: M355MvLogpdf(1)(x, mu, cov, nu)
        ns.pdf = lambda x, nu, mu, cov: pt.exp(ns.logpdf(x, nu, mu, cov))
        return ns
    from .csrc_support import SUPPORTED  # (dist, fn) pairs that have a kernel

    for fn in FN:
        if (dist, fn) in SUPPORTED:
            setattr(ns, fn, (lambda f: lambda x, *p: M355Elemwise(dist, f)(x, *p))(fn))
    ns.sum_logpdf = lambda x, *p: M355SumLogp(dist, "logpdf")(x, *p)
    ns.sum_logcdf = lambda x, *p: M355SumLogp(dist, "logcdf")(x, *p)
    ns.sum_logsf = lambda x, *p: M355SumLogp(dist, "logsf")(x, *p)
    ns.rvs = _rvs(dist)
    return ns
