"""Allocation-free launch helpers over the C-ABI for callers that own their device buffers
(benchmarks, sharded log-likelihood loops): x and out are contiguous torch CUDA tensors, parameters
are Python scalars or float64 CUDA tensors of shape (P,) (row-periodic: x viewed as (n / P, P)).  Work is enqueued on torch's current stream; nothing synchronises."""

from __future__ import annotations

import ctypes

from . import _lib
from ._lib import DIST, F32, F64, FN, I64, Operand


def _dt(t, torch):
    return {torch.float64: F64, torch.float32: F32, torch.int64: I64}[t.dtype]


def _operands(x, params, torch):
    ops = (Operand * (1 + len(params)))()
    ops[0].ptr, ops[0].scalar, ops[0].div, ops[0].mod, ops[0].dtype = x.data_ptr(), 0.0, 1, x.numel(), _dt(x, torch)
    for j, p in enumerate(params, start=1):
        if hasattr(p, "data_ptr"):  # a contiguous float64 CUDA tensor of shape (P,): row-periodic along the flat index
            assert p.is_cuda and p.is_contiguous() and p.dtype == torch.float64 and p.dim() == 1
            ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = p.data_ptr(), 0.0, 1, p.numel(), F64
        else:
            ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = None, float(p), 1, 1, F64
    return ops


def _p(ops, j):
    return ctypes.cast(ctypes.byref(ops, j * ctypes.sizeof(Operand)), ctypes.POINTER(Operand))


def elemwise_workspace(device):
    """A workspace for elemwise_into's launch-time tables (m355_elemwise_workspace_bytes(): 32 MiB).  One per stream
    that may run concurrently with others; calls on one stream may share it."""
    import torch

    return torch.empty(int(_lib.lib().m355_elemwise_workspace_bytes()), dtype=torch.uint8, device=device)


def elemwise_into(dist: str, fn: str, x, params, out, workspace=None) -> None:
    """out[i] = <dist>.<fn>(x[i], *params); `out` may be `x` itself.  With a `workspace` (elemwise_workspace) the
    library may use its launch-time tables (m355_elemwise_ws); without one the exact kernels serve."""
    import torch

    assert x.is_cuda and out.is_cuda and x.is_contiguous() and out.is_contiguous()
    assert out.numel() == x.numel() and out.dtype in (torch.float64, torch.float32)
    ops = _operands(x, params, torch)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):  # the launch goes to HIP's current device: make it x's
        _lib.check(_lib.lib().m355_elemwise_ws(DIST[dist], FN[fn], _dt(out, torch), x.numel(), _p(ops, 0),
                                               len(params), _p(ops, 1), out.data_ptr(),
                                               workspace.data_ptr() if workspace is not None else None,
                                               workspace.numel() if workspace is not None else 0, stream))


def sumlogp_workspace(device):
    import torch

    return torch.empty(int(_lib.lib().m355_sumlogp_workspace_bytes()), dtype=torch.uint8, device=device)


def sumlogp_into(dist: str, fn: str, x, params, out_sum, workspace) -> None:
    """out_sum: float64 CUDA tensor; the total goes to out_sum[0] (pass a slice for other slots)."""
    import torch

    assert x.is_cuda and x.is_contiguous() and out_sum.dtype == torch.float64 and out_sum.is_cuda
    ops = _operands(x, params, torch)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().m355_sumlogp(DIST[dist], FN[fn], x.numel(), _p(ops, 0), len(params), _p(ops, 1),
                                           out_sum.data_ptr(), workspace.data_ptr(), workspace.numel(), stream))


def sumlogp_static_into(dist: str, fn: str, x, params, out_sum) -> None:
    """sumlogp_into through m355_sumlogp_static: no workspace, nothing allocated (partials in the library's static
    device array; calls that overlap on one GPU race on it -- include/m355_dist.h)."""
    import torch

    assert x.is_cuda and x.is_contiguous() and out_sum.dtype == torch.float64 and out_sum.is_cuda
    ops = _operands(x, params, torch)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().m355_sumlogp_static(DIST[dist], FN[fn], x.numel(), _p(ops, 0), len(params), _p(ops, 1),
                                                  out_sum.data_ptr(), stream))


def logpdf_grad_into(dist: str, x, params, sums, dx, workspace) -> None:
    """sums: float64 CUDA tensor of >= 2 + len(params) elements ([sum logpdf, sum d/dx, sum d/dparam_j ...]);
    dx: CUDA tensor like x for the per-element d logpdf/dx, or None."""
    import torch

    assert x.is_cuda and x.is_contiguous() and sums.dtype == torch.float64 and sums.numel() >= 2 + len(params)
    assert dx is None or (dx.is_cuda and dx.is_contiguous() and dx.numel() == x.numel())
    ops = _operands(x, params, torch)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().m355_logpdf_grad(DIST[dist], _dt(dx, torch) if dx is not None else F64, x.numel(),
                                               _p(ops, 0), len(params), _p(ops, 1), sums.data_ptr(),
                                               dx.data_ptr() if dx is not None else None, None, workspace.data_ptr(),
                                               workspace.numel(), stream))


def rvs_into(dist: str, params, out, seed: int, stream_id: int = 0, counter_offset: int = 0) -> None:
    """Fill the contiguous CUDA tensor `out` (float64 / float32, or int64 for the discrete families) with draws of
    `dist` with scalar parameters: Philox counter = counter_offset + i, so a range split over calls or GPUs gives
    the same values as one call."""
    import torch

    assert out.is_cuda and out.is_contiguous()
    ops = (Operand * max(len(params), 1))()
    for j, p in enumerate(params):
        ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = None, float(p), 1, 1, F64
    stream = torch.cuda.current_stream(out.device).cuda_stream
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().m355_rvs(DIST[dist], _dt(out, torch), out.numel(), len(params), _p(ops, 0), int(seed),
                                       int(stream_id), int(counter_offset), out.data_ptr(), stream))
