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
    import pytensor.tensor as pt
    from pytensor.graph.basic import Apply
    from pytensor.graph.op import Op
except ImportError as e:  # pragma: no cover - PyTensor absent here
    raise ImportError("distributions_amd.pytensor_ops needs PyTensor (pip install pytensor); the eager "
                      "functions in distributions_amd.<dist> do not") from e

from . import _dispatch, _mv
from ._lib import FN, NPARAMS


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


def symbolic(dist: str) -> types.SimpleNamespace:
    """Namespace with the reference's function names for one distribution, each building an Op."""
    ns = types.SimpleNamespace()
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
    return ns
