"""A ~150-line stand-in for the slice of PyTensor that distributions_amd/pytensor_ops.py touches, so that the Ops'
make_node (dtype / broadcast / arity logic), perform and L_op bodies EXECUTE on boxes without PyTensor (it cannot be
installed here).  It is test infrastructure, not an emulation of PyTensor's rewriter: variables are
nodes of a plain expression DAG, `.eval()` walks it calling `Op.perform`, `grad()` is reverse mode calling `Op.L_op`.
Where real PyTensor is importable the tests use it instead (tests/test_pytensor_ops.py)."""

import contextlib
import sys
import types

import numpy as np


class TensorType:
    def __init__(self, dtype, shape):
        self.dtype, self.shape, self.ndim = str(np.dtype(dtype)), tuple(shape), len(shape)

    def __call__(self, name=None):
        return Variable(self.dtype, self.shape, name)


class RandomGeneratorType:
    """The type of a variable that holds a numpy Generator (PyTensor: pytensor.tensor.random.type)."""

    dtype, shape, ndim = "object", (), 0

    def __call__(self, name=None):
        v = Variable("float64", (), name)
        v.type = self
        return v


def shared(value, name=None):
    """pytensor.shared: a variable that carries its value (Generators and arrays)."""
    if isinstance(value, np.random.Generator):
        v = RandomGeneratorType()(name)
        v.value = value
        return v
    v = as_tensor_variable(np.asarray(value))
    v.name = name
    return v


class Apply:
    def __init__(self, op, inputs, outputs):
        self.op, self.inputs, self.outputs = op, list(inputs), list(outputs)
        for i, o in enumerate(self.outputs):
            o.owner, o.index = self, i


class Variable:
    def __init__(self, dtype, shape, name=None, value=None):
        self.type = TensorType(dtype, shape)
        self.owner, self.index, self.name, self.value = None, None, name, value

    dtype = property(lambda self: self.type.dtype)
    ndim = property(lambda self: self.type.ndim)

    def __mul__(self, other):
        return _Mul()(self, other)

    __rmul__ = __mul__

    def __add__(self, other):
        return _Add()(self, other)

    def sum(self, axis=None, keepdims=False):
        return _Sum(axis, keepdims)(self)

    def eval(self, givens=None):
        return _evaluate([self], givens or {})[0]


class Op:
    def __call__(self, *inputs):
        node = self.make_node(*inputs)
        return node.outputs[0] if len(node.outputs) == 1 else node.outputs


def as_tensor_variable(v):
    if isinstance(v, Variable):
        return v
    if isinstance(v, (list, tuple)) and any(isinstance(e, Variable) for e in v):
        raise TypeError("the stand-in does not stack variables")
    if isinstance(v, bool):
        a = np.asarray(v)
    elif isinstance(v, int):  # PyTensor: the smallest integer type that holds a Python int
        a = np.asarray(v, dtype=np.int8 if -128 <= v < 128 else np.int64)
    else:
        a = np.asarray(v)  # Python floats -> float64 (floatX)
    return Variable(a.dtype, a.shape, value=a)


constant = as_tensor_variable


def tensor(dtype, shape, name=None):
    return Variable(dtype, shape, name=name)


def _bshape(a, b):
    return tuple(np.broadcast_shapes(tuple(1 if d is None else d for d in a), tuple(1 if d is None else d for d in b)))


class _Binary(Op):
    def make_node(self, a, b):
        a, b = as_tensor_variable(a), as_tensor_variable(b)
        nd = max(a.ndim, b.ndim)
        pad = lambda v: (1,) * (nd - v.ndim) + v.type.shape
        shape = tuple(x if y in (1, None) and x is not None else (y if x in (1, None) else x) for x, y in zip(pad(a), pad(b)))
        return Apply(self, [a, b], [Variable(np.result_type(a.dtype, b.dtype), shape)])


class _Mul(_Binary):
    def perform(self, node, inputs, out):
        out[0][0] = np.asarray(inputs[0] * inputs[1])

    def L_op(self, inputs, outputs, gs):
        return [_sum_to_shape(gs[0] * inputs[1], inputs[0]), _sum_to_shape(gs[0] * inputs[0], inputs[1])]


class _Add(_Binary):
    def perform(self, node, inputs, out):
        out[0][0] = np.asarray(inputs[0] + inputs[1])

    def L_op(self, inputs, outputs, gs):
        return [_sum_to_shape(gs[0], inputs[0]), _sum_to_shape(gs[0], inputs[1])]


class _SumToShapeOf(Op):
    def make_node(self, g, like):
        return Apply(self, [g, like], [Variable(g.dtype, like.type.shape)])

    def perform(self, node, inputs, out):
        g, like = inputs
        lead = g.ndim - like.ndim
        if lead:
            g = g.sum(axis=tuple(range(lead)))
        axes = tuple(a for a in range(like.ndim) if like.shape[a] == 1 and g.shape[a] != 1)
        out[0][0] = g.sum(axis=axes, keepdims=True) if axes else g


def _sum_to_shape(g, like):
    return _SumToShapeOf()(g, like)


class _Sum(Op):
    def __init__(self, axis, keepdims):
        self.axis = (axis,) if isinstance(axis, int) else (tuple(axis) if axis is not None else None)
        self.keepdims = keepdims

    def make_node(self, a):
        ax = range(a.ndim) if self.axis is None else self.axis
        shape = tuple((1 if i in ax else d) for i, d in enumerate(a.type.shape)) if self.keepdims else tuple(
            d for i, d in enumerate(a.type.shape) if i not in ax)
        return Apply(self, [a], [Variable(a.dtype, shape)])

    def perform(self, node, inputs, out):
        out[0][0] = np.asarray(inputs[0].sum(axis=self.axis, keepdims=self.keepdims))

    def L_op(self, inputs, outputs, gs):
        return [_BroadcastLike(self.axis, self.keepdims)(gs[0], inputs[0])]


class _BroadcastLike(Op):
    def __init__(self, axis, keepdims):
        self.axis, self.keepdims = axis, keepdims

    def make_node(self, g, like):
        return Apply(self, [g, like], [Variable(g.dtype, like.type.shape)])

    def perform(self, node, inputs, out):
        g, like = inputs
        if not self.keepdims and self.axis is not None:
            g = np.expand_dims(g, self.axis)
        out[0][0] = np.broadcast_to(g, like.shape).copy()


class _Exp(Op):
    def make_node(self, a):
        return Apply(self, [a], [Variable(a.dtype, a.type.shape)])

    def perform(self, node, inputs, out):
        out[0][0] = np.exp(inputs[0])


def _evaluate(outputs, givens):
    cache = {}

    def val(v):
        if id(v) in cache:
            return cache[id(v)]
        if v in givens:
            r = givens[v] if isinstance(v.type, RandomGeneratorType) else np.asarray(givens[v], dtype=v.dtype)
        elif v.owner is None:
            if v.value is None:
                raise ValueError(f"no value for input {v.name}")
            r = v.value
        else:
            node = v.owner
            storage = [[None] for _ in node.outputs]
            node.op.perform(node, [val(i) for i in node.inputs], storage)
            for o, st in zip(node.outputs, storage):
                cache[id(o)] = st[0]
            r = cache[id(v)]
        cache[id(v)] = r
        return r

    return [val(o) for o in outputs]


def grad(cost, wrt):
    """Reverse mode over the expression DAG through Op.L_op (which receives and returns symbolic variables)."""
    order, seen = [], set()

    def topo(v):
        if id(v) in seen or v.owner is None:
            return
        seen.add(id(v))
        for i in v.owner.inputs:
            topo(i)
        order.append(v.owner)

    topo(cost)
    g = {id(cost): as_tensor_variable(np.ones(tuple(1 if d is None else d for d in cost.type.shape), dtype=cost.dtype))}
    done = set()
    for node in reversed(order):
        if id(node) in done:
            continue
        done.add(id(node))
        gs = [g.get(id(o)) for o in node.outputs]
        if all(x is None for x in gs) or not hasattr(node.op, "L_op"):
            continue
        gs = [x if x is not None else as_tensor_variable(np.zeros(())) for x in gs]
        for i, gi in zip(node.inputs, node.op.L_op(node.inputs, node.outputs, gs)):
            g[id(i)] = gi if id(i) not in g else g[id(i)] + gi
    return [g[id(w)] for w in wrt]


def function(inputs, outputs):
    return lambda *vals: _evaluate(outputs, dict(zip(inputs, vals)))


@contextlib.contextmanager
def installed():
    """sys.modules entries `pytensor`, `pytensor.tensor`, `pytensor.graph.basic`, `pytensor.graph.op` for the duration
    of the block; distributions_amd.pytensor_ops is (re-)imported against them and removed again afterwards."""
    pt = types.ModuleType("pytensor.tensor")
    pt.as_tensor_variable, pt.constant, pt.tensor, pt.exp = as_tensor_variable, constant, tensor, (lambda a: _Exp()(a))
    pt.dscalar = lambda name=None: Variable("float64", (), name)
    pt.dvector = lambda name=None: Variable("float64", (None,), name)
    pt.dmatrix = lambda name=None: Variable("float64", (None, None), name)
    pt.fmatrix = lambda name=None: Variable("float32", (None, None), name)
    pt.lvector = lambda name=None: Variable("int64", (None,), name)
    pt.dtensor3 = lambda name=None: Variable("float64", (None, None, None), name)
    pt.broadcast_shape = lambda *shapes, arrays_are_shapes=False: tuple(np.broadcast_shapes(*shapes))
    root = types.ModuleType("pytensor")
    root.tensor, root.grad, root.function, root.shared = pt, grad, function, shared
    graph = types.ModuleType("pytensor.graph")
    basic, opm = types.ModuleType("pytensor.graph.basic"), types.ModuleType("pytensor.graph.op")
    basic.Apply, opm.Op = Apply, Op
    graph.basic, graph.op, root.graph = basic, opm, graph
    names = {"pytensor": root, "pytensor.tensor": pt, "pytensor.graph": graph, "pytensor.graph.basic": basic,
             "pytensor.graph.op": opm}
    saved = {k: sys.modules.get(k) for k in list(names) + ["distributions_amd.pytensor_ops"]}
    sys.modules.update(names)
    sys.modules.pop("distributions_amd.pytensor_ops", None)
    try:
        import importlib

        yield importlib.import_module("distributions_amd.pytensor_ops"), root
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        import distributions_amd

        if saved["distributions_amd.pytensor_ops"] is None and hasattr(distributions_amd, "pytensor_ops"):
            delattr(distributions_amd, "pytensor_ops")
