"""Host side of categorical / multinomial: shape logic of categorical.py:60-117 and
multinomial.py:35-58 of the reference, then one call into the C-ABI (csrc/m355_catmult.hip)."""

from __future__ import annotations

import ctypes
from math import prod

import numpy as np

from . import _lib
from ._dispatch import _device_of, _is_torch, _seed_of, _shape_of, _torch, _x_dtype
from ._lib import F32, F64, FN, I64

_vp, _i32, _i64, _dbl, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_size_t
_u64, _u32 = ctypes.c_uint64, ctypes.c_uint32
_bound = False
CATEGORICAL_STREAM, MULTINOMIAL_STREAM = 1000, 1001  # Philox stream ids (elementwise families use their dist id)


def _bind():
    global _bound
    L = _lib.lib()
    if not _bound:
        L.m355_categorical_workspace_bytes.restype = _sz
        L.m355_categorical_workspace_bytes.argtypes = [_i32]
        L.m355_categorical.restype = _i32
        L.m355_categorical.argtypes = [_i32, _i64, _vp, _i32, _vp, _i32, _vp, _vp, _sz, _vp]
        L.m355_categorical_rvs.restype = _i32
        L.m355_categorical_rvs.argtypes = [_i64, _vp, _i32, _u64, _u32, _u64, _vp, _vp, _sz, _vp]
        L.m355_multinomial_logpdf.restype = _i32
        L.m355_multinomial_logpdf.argtypes = [_i64, _i32, _vp, _i32, _vp, _dbl, _i64, _vp, _i64, _vp, _vp]
        L.m355_multinomial_rvs.restype = _i32
        L.m355_multinomial_rvs.argtypes = [_i64, _i32, _dbl, _vp, _u64, _u32, _u64, _vp, _vp]
        _bound = True
    return L


def _as_tensor(a, torch, device, dtype):
    if _is_torch(a):
        return a.to(device=device, dtype=dtype)
    return torch.from_numpy(np.array(np.asarray(a), order="C", copy=True)).to(device=device, dtype=dtype)


def _arr(a):
    """Lists / tuples / Python scalars become NumPy arrays (so they have a shape and a dtype)."""
    return a if _is_torch(a) or isinstance(a, np.ndarray) else np.asarray(a)


def _code(dt, torch):
    return {torch.float64: F64, torch.float32: F32, torch.int64: I64}[dt]


def _p_vector(p, torch, device):
    if len(_shape_of(p)) != 1:
        # categorical.py:73 gathers log(p)[x] along the FIRST axis: only a single probability
        # vector gives a per-category density in the reference, so that is what the path takes
        raise ValueError("categorical: p must be one probability vector of shape (k,)")
    return _as_tensor(p, torch, device, torch.float64).contiguous()


def categorical(fn: str, x, p):
    torch = _torch()
    L = _bind()
    x, p = _arr(x), _arr(p)
    device = _device_of((x, p), torch)
    want_torch = _is_torch(x) or _is_torch(p)
    shape = _shape_of(x)
    n = prod(shape)
    with torch.cuda.device(device):
        pt_ = _p_vector(p, torch, device)
        k = pt_.numel()
        xdt = _x_dtype(x, torch, False)
        xt = _as_tensor(x, torch, device, xdt).contiguous()
        out = torch.empty(n, dtype=torch.float64, device=device)
        if n:
            ws_bytes = int(L.m355_categorical_workspace_bytes(k))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(L.m355_categorical(FN[fn], n, xt.data_ptr(), _code(xdt, torch), pt_.data_ptr(), k,
                                          out.data_ptr(), ws.data_ptr(), ws_bytes, stream))
    out = out.reshape(shape)
    return out if want_torch else out.cpu().numpy()


def categorical_rvs(p, size=None, random_state=None):
    torch = _torch()
    L = _bind()
    p = _arr(p)
    device = _device_of((p,), torch)
    shape = () if size is None else ((int(size),) if isinstance(size, (int, np.integer)) else tuple(int(s) for s in size))
    n = prod(shape)
    seed, offset = _seed_of(random_state)
    with torch.cuda.device(device):
        pt_ = _p_vector(p, torch, device)
        k = pt_.numel()
        out = torch.empty(n, dtype=torch.int64, device=device)
        if n:
            ws_bytes = int(L.m355_categorical_workspace_bytes(k))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(L.m355_categorical_rvs(n, pt_.data_ptr(), k, seed, CATEGORICAL_STREAM, offset, out.data_ptr(),
                                              ws.data_ptr(), ws_bytes, stream))
    out = out.reshape(shape)
    return out if _is_torch(p) else out.cpu().numpy()


def multinomial_logpdf(x, n, p):
    torch = _torch()
    L = _bind()
    x, n, p = _arr(x), _arr(n), _arr(p)
    device = _device_of((x, n, p), torch)
    want_torch = any(_is_torch(a) for a in (x, n, p))
    xs, ns, ps = _shape_of(x), _shape_of(n), _shape_of(p)
    if len(xs) < 1 or len(ps) < 1 or xs[-1] != ps[-1]:
        raise ValueError(f"multinomial: x (..., k) and p (..., k) must share the last axis, got {xs} and {ps}")
    k = xs[-1]
    batch = tuple(np.broadcast_shapes(xs[:-1], ns, ps[:-1]))
    rows = prod(batch)
    with torch.cuda.device(device):
        xdt = _x_dtype(x, torch, False)
        xt = _as_tensor(x, torch, device, xdt).expand(batch + (k,)).contiguous()
        pt_ = _as_tensor(p, torch, device, torch.float64)
        if prod(ps[:-1]) == 1:
            pt_, p_rows = pt_.reshape(k).contiguous(), 1
        else:
            pt_, p_rows = pt_.expand(batch + (k,)).contiguous(), rows
        if prod(ns) == 1:
            n_ptr, n_scalar, n_mod, keep = None, float(np.asarray(n.cpu() if _is_torch(n) else n).reshape(-1)[0]), 1, None
        else:
            keep = _as_tensor(n, torch, device, torch.float64).expand(batch).contiguous()
            n_ptr, n_scalar, n_mod = keep.data_ptr(), 0.0, rows
        out = torch.empty(rows, dtype=torch.float64, device=device)
        if rows:
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(L.m355_multinomial_logpdf(rows, k, xt.data_ptr(), _code(xdt, torch), n_ptr, n_scalar, n_mod,
                                                 pt_.data_ptr(), p_rows, out.data_ptr(), stream))
        del keep
    out = out.reshape(batch)
    return out if want_torch else out.cpu().numpy()


def multinomial_rvs(n, p, size=None, random_state=None):
    torch = _torch()
    L = _bind()
    n, p = _arr(n), _arr(p)
    device = _device_of((p,), torch)
    if len(_shape_of(p)) != 1 or _shape_of(n) != ():
        raise ValueError("multinomial.rvs: scalar n and one probability vector p of shape (k,)")
    shape = () if size is None else ((int(size),) if isinstance(size, (int, np.integer)) else tuple(int(s) for s in size))
    rows = prod(shape)
    seed, offset = _seed_of(random_state)
    with torch.cuda.device(device):
        pt_ = _as_tensor(p, torch, device, torch.float64).contiguous()
        k = pt_.numel()
        out = torch.empty(rows * k, dtype=torch.int64, device=device)
        if rows:
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(L.m355_multinomial_rvs(rows, k, float(n), pt_.data_ptr(), seed, MULTINOMIAL_STREAM, offset,
                                              out.data_ptr(), stream))
    out = out.reshape(shape + (k,))
    return out if _is_torch(p) else out.cpu().numpy()
