"""Host side of the elementwise path: what PyTensor does symbolically around the fused Elemwise
loop (NumPy-style broadcasting of (x, *params), dtype promotion, 0-d handling), done eagerly, then
one call into the C-ABI.  `plan()` is pure Python (no GPU, no library) and is unit-tested on CPU.
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from math import prod
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import DIST, F32, F64, FN, I64, NPARAMS, STAT, Operand


# ------------------------------------------------------------------------------------------ plan
@dataclass
class OperandPlan:
    kind: str  # "scalar" | "full" | "inner" | "outer" | "middle" | "expand"
    div: int = 1
    mod: int = 1


@dataclass
class Plan:
    shape: Tuple[int, ...]
    n: int
    out_f32: bool
    operands: List[OperandPlan] = field(default_factory=list)


def _shape_of(a: Any) -> Tuple[int, ...]:
    s = getattr(a, "shape", None)
    return tuple(int(d) for d in s) if s is not None else ()


def _dtype_name(a: Any) -> Optional[str]:
    """'float32', 'int64', ... for arrays/tensors; None for Python scalars (weak)."""
    dt = getattr(a, "dtype", None)
    if dt is None:
        return None
    return str(dt).replace("torch.", "")


def classify(op_shape: Sequence[int], out_shape: Sequence[int]) -> OperandPlan:
    """How a broadcast operand is indexed by the flat output index i: (i // div) % mod."""
    n = prod(out_shape)
    size = prod(op_shape)
    if size == 1:
        return OperandPlan("scalar")
    nd = len(out_shape)
    s = (1,) * (nd - len(op_shape)) + tuple(op_shape)
    if s == tuple(out_shape):
        return OperandPlan("full", 1, n)
    nz = [d for d in range(nd) if out_shape[d] != 1]  # dimensions that matter
    match = [s[d] == out_shape[d] for d in nz]
    first = match.index(True)
    last = len(match) - 1 - match[::-1].index(True)
    if all(match[first : last + 1]):
        # one contiguous run of non-broadcast dimensions: index = (i // prod(trailing)) % size
        div = prod(out_shape[d] for d in nz[last + 1 :])
        kind = "inner" if div == 1 else ("outer" if first == 0 else "middle")
        return OperandPlan(kind, div, size)
    return OperandPlan("expand", 1, n)


def plan(x: Any, params: Sequence[Any]) -> Plan:
    """Broadcast shape, output dtype and per-operand index plan for (x, *params)."""
    shapes = [_shape_of(a) for a in (x, *params)]
    shape = tuple(np.broadcast_shapes(*shapes))
    names = [_dtype_name(a) for a in (x, *params)]
    floats = [nm for nm in names if nm is not None and nm.startswith(("float", "bfloat"))]
    # float32 only if every floating *array* input is <= 32 bits (Python scalars are weak) and at
    # least one exists; integer-only inputs give float64 (PyTensor upcasts int64 -> float64).
    out_f32 = bool(floats) and all(nm in ("float32", "float16", "bfloat16") for nm in floats)
    ints = [nm for nm in names if nm is not None and nm.startswith(("int", "uint")) and nm not in ("int8", "int16", "uint8")]
    if ints:
        out_f32 = False
    return Plan(shape, prod(shape), out_f32, [classify(s, shape) for s in shapes])


# ---------------------------------------------------------------------------------------- launch
def _torch():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "distributions_amd needs a GPU: torch.cuda.is_available() is False and there is no CPU fallback"
        )
    return torch


def _is_torch(a: Any) -> bool:
    return type(a).__module__.startswith("torch")


def _to_device(a: Any, torch, device, dtype):
    if _is_torch(a):
        t = a
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
    return t.to(device=device, dtype=dtype, non_blocking=False)


def _x_dtype(a: Any, torch, out_f32: bool):
    nm = _dtype_name(a)
    if nm is None:
        return torch.float64
    if nm.startswith(("int", "uint", "bool")):
        return torch.int64
    if nm == "float64":
        return torch.float64
    return torch.float32 if out_f32 else torch.float64


def _scalar_value(a: Any) -> float:
    if _is_torch(a):
        return float(a.reshape(-1)[0].item())
    return float(np.asarray(a).reshape(-1)[0])


def _device_of(args, torch):
    for a in args:
        if _is_torch(a) and a.is_cuda:
            return a.device
    return torch.device("cuda", torch.cuda.current_device())


def _build_operands(pl: Plan, args: Sequence[Any], torch, device):
    """Operand structs + the tensors that must stay alive until the launch is enqueued."""
    keep = []
    ops = (Operand * len(args))()
    for j, (a, op) in enumerate(zip(args, pl.operands)):
        o = ops[j]
        o.reserved = 0
        if op.kind == "scalar":
            o.ptr, o.scalar, o.div, o.mod, o.dtype = None, _scalar_value(a), 1, 1, F64
            continue
        dt = _x_dtype(a, torch, pl.out_f32) if j == 0 else torch.float64
        t = _to_device(a, torch, device, dt)
        if op.kind == "expand":
            t = t.expand(pl.shape)
        t = t.contiguous()
        keep.append(t)
        o.ptr, o.scalar, o.div, o.mod = t.data_ptr(), 0.0, op.div, op.mod
        o.dtype = {torch.float64: F64, torch.float32: F32, torch.int64: I64}[dt]
    return ops, keep


# families whose logcdf / logsf / cdf / sf / ppf / isf the library can serve from launch-time tables, given a workspace
# (include/m355_dist.h, m355_elemwise_ws): 20 KiB for scalar parameters; 9.2 KiB per column + one bit per element for
# row-periodic ones
TABLE_DISTS = ("gamma", "beta", "studentt", "halfstudentt", "chisquared", "inversegamma", "scaledinversechisquared",
               "betascaled")
# discrete families whose cdf / sf / logcdf / logsf with row-periodic parameters are looked up in per-column windows of
# exact values (k_dcol: 2 + 2048 doubles per column), from 8 x the table threshold (2**24 elements by default)
DTABLE_DISTS = ("poisson", "binomial", "negativebinomial", "betabinomial", "hypergeometric", "zi_poisson", "zi_binomial",
                "zi_negativebinomial")
_TABLE_WS_MAX = 1 << 30


def _table_workspace(dist: str, fn: str, pl: Plan, torch, device):
    """Scratch for m355_elemwise_ws, from torch's caching allocator (a per-call block: calls on different streams never
    share one), or None where no table can apply."""
    if pl.operands[0].kind != "full":
        return None
    period = 0
    for op in pl.operands[1:]:
        if op.kind == "scalar":
            continue
        if op.div != 1 or (period and op.mod != period) or op.mod >= pl.n:
            return None  # not row-periodic: the library's generic kernel
        period = int(op.mod)
    densities = fn in ("logpdf", "pdf")
    if period and dist in DTABLE_DISTS and fn in ("cdf", "sf", "logcdf", "logsf", "ppf", "isf") and pl.n >= 8 * int(_lib.lib().m355_tab_min_n()):
        need = int(_lib.lib().m355_elemwise_workspace_bytes_for(pl.n, period))
        return torch.empty(need, dtype=torch.uint8, device=device) if 0 < need <= _TABLE_WS_MAX else None
    if period == 0 and (densities or dist not in TABLE_DISTS):
        return None
    if period and (densities or dist not in TABLE_DISTS):
        need = period * 16 * 8  # the derived constants of the P parameter sets (k_periodic_fixed)
    else:
        need = int(_lib.lib().m355_elemwise_workspace_bytes_for(pl.n, period))
    if need == 0 or need > _TABLE_WS_MAX:
        return None
    return torch.empty(need, dtype=torch.uint8, device=device)


def _wants_torch(args) -> bool:
    return any(_is_torch(a) for a in args)


def elemwise(dist: str, fn: str, x: Any, *params: Any):
    """<dist>.<fn>(x, *params) on the GPU.  Returns a torch CUDA tensor if any input was a torch
    tensor, else a NumPy array (0-d inputs give a 0-d result), dtype float64 or float32."""
    if len(params) != NPARAMS[dist]:
        raise TypeError(f"{dist}.{fn} takes {NPARAMS[dist]} parameters, got {len(params)}")
    torch = _torch()
    L = _lib.lib()
    args = (x, *params)
    pl = plan(x, params)
    device = _device_of(args, torch)
    out_dt = torch.float32 if pl.out_f32 else torch.float64
    with torch.cuda.device(device):
        out = torch.empty(pl.n, dtype=out_dt, device=device)
        if pl.n:
            ops, keep = _build_operands(pl, args, torch, device)
            stream = torch.cuda.current_stream(device).cuda_stream
            ws = _table_workspace(dist, fn, pl, torch, device)
            rc = L.m355_elemwise_ws(DIST[dist], FN[fn], F32 if pl.out_f32 else F64, pl.n,
                                    ctypes.byref(ops[0]), len(params),
                                    ctypes.cast(ctypes.byref(ops, ctypes.sizeof(Operand)), ctypes.POINTER(Operand)),
                                    out.data_ptr(), ws.data_ptr() if ws is not None else None,
                                    ws.numel() if ws is not None else 0, stream)
            _lib.check(rc)
            del keep, ws  # (torch's allocator is stream-ordered: the block is not handed out again before this launch ran)
    out = out.reshape(pl.shape)
    if _wants_torch(args):
        return out
    return out.cpu().numpy()


def sumlogp(dist: str, fn: str, x: Any, *params: Any):
    """sum(<dist>.<fn>(x, *params)) in float64 without materialising the elementwise result
    (the reference spelling is pt.sum(dist.logpdf(...))).  Returns a 0-d torch
    tensor on the device if any input was a torch tensor, else a Python float."""
    if len(params) != NPARAMS[dist]:
        raise TypeError(f"{dist}.{fn} takes {NPARAMS[dist]} parameters, got {len(params)}")
    torch = _torch()
    L = _lib.lib()
    args = (x, *params)
    pl = plan(x, params)
    device = _device_of(args, torch)
    with torch.cuda.device(device):
        out = torch.empty(1, dtype=torch.float64, device=device)
        ws_bytes = int(L.m355_sumlogp_workspace_bytes())
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        ops, keep = _build_operands(pl, args, torch, device)
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = L.m355_sumlogp(DIST[dist], FN[fn], pl.n, ctypes.byref(ops[0]), len(params),
                            ctypes.cast(ctypes.byref(ops, ctypes.sizeof(Operand)), ctypes.POINTER(Operand)),
                            out.data_ptr(), ws.data_ptr(), ws_bytes, stream)
        _lib.check(rc)
        del keep
    if _wants_torch(args):
        return out.reshape(())
    return float(out.item())


# -------------------------------------------------------------------------------------- gradient
def _unbroadcast(t, shape, torch):
    """Sum a tensor of the broadcast shape down to an operand's shape (what PyTensor's Elemwise L_op
    does with the gradient of a broadcast input)."""
    shape = tuple(shape)
    if tuple(t.shape) == shape:
        return t
    lead = t.dim() - len(shape)
    if lead:
        t = t.sum(dim=tuple(range(lead)))
    axes = tuple(i for i, (a, b) in enumerate(zip(t.shape, shape)) if b == 1 and a != 1)
    if axes:
        t = t.sum(dim=axes, keepdim=True)
    return t.reshape(shape)


def logpdf_grad(dist: str, x: Any, *params: Any, elementwise: bool = False, x_grad: bool = True):
    """Value and gradient of sum(<dist>.logpdf(x, *params)) in one pass over x (m355_logpdf_grad; what
    pytensor.grad of the reference's graph gives).

    elementwise=False: returns (total, [g_x, g_param_0, ...]) where every gradient has the shape of its
    argument (scalar arguments get the summed partial straight from the kernel's reduction; array
    arguments get the per-element partials, summed over the axes they were broadcast along).
    elementwise=True: returns (total, [d logpdf/dx, d logpdf/dparam_0, ...]) per element of the
    broadcast shape (the factors an Elemwise L_op multiplies with the output gradient).
    x_grad=False: x is data, its gradient is returned as None and, for an array x with scalar parameters,
    never computed (the kernel then costs little more than the value alone)."""
    if len(params) != NPARAMS[dist]:
        raise TypeError(f"{dist}.logpdf takes {NPARAMS[dist]} parameters, got {len(params)}")
    if dist not in _lib.GRAD:
        raise NotImplementedError(f"{dist} has no gradient kernel")
    torch = _torch()
    L = _lib.lib()
    args = (x, *params)
    pl = plan(x, params)
    device = _device_of(args, torch)
    out_dt = torch.float32 if pl.out_f32 else torch.float64
    np_ = len(params)
    with torch.cuda.device(device):
        sums = torch.zeros(2 + np_, dtype=torch.float64, device=device)
        ws_bytes = int(L.m355_sumlogp_workspace_bytes())
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
        want = [elementwise or op.kind != "scalar" for op in pl.operands]
        want[0] = want[0] and x_grad
        bufs = [torch.empty(pl.n, dtype=out_dt, device=device) if (w and pl.n) else None for w in want]
        dptrs = (ctypes.c_void_p * max(np_, 1))(*[b.data_ptr() if b is not None else None for b in bufs[1:]])
        ops, keep = _build_operands(pl, args, torch, device)
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = L.m355_logpdf_grad(DIST[dist], F32 if pl.out_f32 else F64, pl.n, ctypes.byref(ops[0]), np_,
                                ctypes.cast(ctypes.byref(ops, ctypes.sizeof(Operand)), ctypes.POINTER(Operand)),
                                sums.data_ptr(), bufs[0].data_ptr() if bufs[0] is not None else None, dptrs,
                                ws.data_ptr(), ws_bytes, stream)
        _lib.check(rc)
        del keep
        grads = []
        for j, (a, b) in enumerate(zip(args, bufs)):
            if j == 0 and not x_grad:
                grads.append(None)
                continue
            if elementwise:
                g = b.reshape(pl.shape) if b is not None else torch.zeros(pl.shape, dtype=out_dt, device=device)
            elif b is None and pl.operands[j].kind != "scalar":  # empty broadcast shape
                g = torch.zeros(_shape_of(a), dtype=out_dt, device=device)
            elif b is None:
                g = sums[1 + j].to(out_dt).reshape(_shape_of(a))
            else:
                g = _unbroadcast(b.reshape(pl.shape), _shape_of(a), torch)
            grads.append(g)
    if _wants_torch(args):
        return sums[0].reshape(()), grads
    host = sums.cpu().numpy()
    return float(host[0]), [g.cpu().numpy() if g is not None else None for g in grads]


# --------------------------------------------------------------------------------------- moments
def moment(dist: str, stat: str, *params: Any):
    """<dist>.<stat>(*params) for stat in mean / mode / median / var / std / skewness / kurtosis / entropy
    (m355_moment): one value per parameter set of the broadcast shape of the parameters.  float64, or float32
    when every floating array parameter is float32; torch tensor out if any parameter is a torch tensor."""
    if len(params) != NPARAMS[dist]:
        raise TypeError(f"{dist}.{stat} takes {NPARAMS[dist]} parameters, got {len(params)}")
    torch = _torch()
    L = _lib.lib()
    pl = plan(0.0, params)
    device = _device_of(params, torch)
    out_dt = torch.float32 if pl.out_f32 else torch.float64
    with torch.cuda.device(device):
        out = torch.empty(pl.n, dtype=out_dt, device=device)
        if pl.n:
            ops, keep = _build_operands(pl, (0.0, *params), torch, device)
            stream = torch.cuda.current_stream(device).cuda_stream
            rc = L.m355_moment(DIST[dist], STAT[stat], F32 if pl.out_f32 else F64, pl.n, len(params),
                               ctypes.cast(ctypes.byref(ops, ctypes.sizeof(Operand)), ctypes.POINTER(Operand)),
                               out.data_ptr(), stream)
            _lib.check(rc)
            del keep
    out = out.reshape(pl.shape)
    if dist == "discreteweibull" and stat == "mode":  # pt.cast(..., "int64"), discreteweibull.py:19
        out = out.to(torch.int64)
    if _wants_torch(params):
        return out
    return out.cpu().numpy()


# ------------------------------------------------------------------------------------------- rvs
_STREAM_ID = DIST  # Philox stream id = distribution id


def _seed_of(random_state) -> Tuple[int, int]:
    """(seed, counter offset) from the reference's `random_state` argument.  None: fresh OS entropy;
    int: that seed; numpy Generator: a 63-bit seed drawn from it (which advances it, so repeated
    calls differ, like the next_rng threading of pt.random); (seed, offset) tuple: explicit, for
    sharded or resumed generation."""
    if random_state is None:
        return int.from_bytes(__import__("os").urandom(8), "little") >> 1, 0
    if isinstance(random_state, tuple):
        return int(random_state[0]), int(random_state[1])
    if isinstance(random_state, (int, np.integer)):
        return int(random_state), 0
    if hasattr(random_state, "integers"):
        return int(random_state.integers(0, 2**63 - 1)), 0
    raise TypeError(f"random_state of type {type(random_state).__name__} is not supported")


def rvs(dist: str, params: Sequence[Any], size=None, random_state=None, as_torch: bool = False):
    """<dist>.rvs(*params, size=None, random_state=None): Philox4x32-10 counter-based draws on the
    GPU (csrc/m355_rng.h).  size None -> broadcast shape of the parameters; int or tuple otherwise
    (parameters must broadcast to it).  int64 for poisson/binomial, float64 otherwise."""
    if len(params) != NPARAMS[dist]:
        raise TypeError(f"{dist}.rvs takes {NPARAMS[dist]} parameters, got {len(params)}")
    torch = _torch()
    L = _lib.lib()
    pshape = tuple(np.broadcast_shapes(*[_shape_of(a) for a in params]))
    if size is None:
        shape = pshape
    else:
        shape = (int(size),) if isinstance(size, (int, np.integer)) else tuple(int(s) for s in size)
        if tuple(np.broadcast_shapes(shape, pshape)) != shape:
            raise ValueError(f"parameters of shape {pshape} do not broadcast to size {shape}")
    n = prod(shape)
    seed, offset = _seed_of(random_state)
    discrete = dist in _lib.DISCRETE
    device = _device_of(params, torch)
    pl = Plan(shape, n, False, [OperandPlan("scalar")] + [classify(_shape_of(a), shape) for a in params])
    with torch.cuda.device(device):
        out = torch.empty(n, dtype=torch.int64 if discrete else torch.float64, device=device)
        if n:
            ops, keep = _build_operands(pl, (0.0, *params), torch, device)
            stream = torch.cuda.current_stream(device).cuda_stream
            rc = L.m355_rvs(DIST[dist], I64 if discrete else F64, n, len(params),
                            ctypes.cast(ctypes.byref(ops, ctypes.sizeof(Operand)), ctypes.POINTER(Operand)),
                            seed, _STREAM_ID[dist], offset, out.data_ptr(), stream)
            _lib.check(rc)
            del keep
    out = out.reshape(shape)
    if as_torch or _wants_torch(params):
        return out
    return out.cpu().numpy()
