"""m355_sumlogp_static -- the sum-logp entry point that allocates nothing (include/m355_dist.h): called through the
C-ABI and checked against m355_sumlogp (scalar parameters: same kernels and reduction order, bit-identical; row-periodic
parameters: the parity bar) and against the oracle, on two calls in a row, the second with a smaller input (the static
array keeps the first call's partials)."""

import ctypes
import math

import numpy as np
import pytest

from oracle import restatement as R

pytestmark = pytest.mark.gpu


def _ops(lib_mod, x, params):
    from distributions_amd._lib import F64, Operand

    ops = (Operand * (1 + len(params)))()
    ops[0].ptr, ops[0].scalar, ops[0].div, ops[0].mod, ops[0].dtype = x.data_ptr(), 0.0, 1, x.numel(), F64
    for j, p in enumerate(params, start=1):
        ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = None, float(p), 1, 1, F64
    at = lambda j: ctypes.cast(ctypes.byref(ops, j * ctypes.sizeof(Operand)), ctypes.POINTER(Operand))
    return ops, at(0), at(1)


def test_sumlogp_static_matches_workspace_entry_point(gpu):
    import torch

    from distributions_amd import _lib
    from distributions_amd._lib import DIST, FN

    L = _lib.lib()
    rng = np.random.default_rng(77)
    ws = torch.empty(int(L.m355_sumlogp_workspace_bytes()), dtype=torch.uint8, device=gpu)
    out = torch.full((2,), float("nan"), dtype=torch.float64, device=gpu)
    stream = torch.cuda.current_stream(gpu).cuda_stream
    cases = (
        ("normal", "logpdf", (0.3, 1.7), lambda n: rng.normal(0, 1, n)),
        ("gamma", "logpdf", (2.5, 1.3), lambda n: rng.gamma(2.5, 1 / 1.3, n)),
        ("beta", "logpdf", (2.0, 5.0), lambda n: rng.beta(2, 5, n)),
        ("gamma", "logcdf", (2.5, 1.3), lambda n: rng.gamma(2.5, 1 / 1.3, n)),
    )
    # first call: 4096 workgroups' worth of partials; second: a smaller input that uses fewer slots of the array
    for name, fn, params, draw in cases:
        for n in (5_000_003, 70_001, 0):
            xh = draw(n)
            x = torch.from_numpy(xh).to(gpu)
            keep, px, pp = _ops(_lib, x, params)
            out.fill_(float("nan"))
            _lib.check(L.m355_sumlogp(DIST[name], FN[fn], n, px, len(params), pp, out.data_ptr(), ws.data_ptr(),
                                      ws.numel(), stream))
            _lib.check(L.m355_sumlogp_static(DIST[name], FN[fn], n, px, len(params), pp, out.data_ptr() + 8, stream))
            a, b = (float(v) for v in out.cpu())
            assert a == b, (name, fn, n, a, b)  # bit-identical to the workspace entry point
            terms = R.TABLE[f"{name}_{fn}"](xh, *params)
            assert abs(b - math.fsum(terms)) <= 1e-12 * float(np.sum(np.abs(terms))), (name, fn, n)
            del keep


def test_sumlogp_static_row_periodic_parameters(gpu):
    """(P,) parameters: the static array has no room for m355_sumlogp's table of 1024 x 3 derived constants, so the
    static path keeps k_periodic (LDS table) where the workspace path takes k_derive_table + k_periodic_prod: the same
    total within 1e-12 * sum |term| (include/m355_dist.h promises no bit identity here); two calls, the second smaller."""
    import torch

    from distributions_amd import _lib
    from distributions_amd._lib import DIST, F64, FN, Operand

    L = _lib.lib()
    rng = np.random.default_rng(78)
    P = 1024
    ws = torch.empty(int(L.m355_sumlogp_workspace_bytes()), dtype=torch.uint8, device=gpu)
    out = torch.full((2,), float("nan"), dtype=torch.float64, device=gpu)
    stream = torch.cuda.current_stream(gpu).cuda_stream
    al, be = rng.uniform(1.0, 6.0, P), rng.uniform(0.5, 3.0, P)
    ald, bed = torch.from_numpy(al).to(gpu), torch.from_numpy(be).to(gpu)
    for fn in ("logpdf", "logcdf"):
        for rows in (3000, 40):
            xh = rng.gamma(al, 1 / be, (rows, P))
            x = torch.from_numpy(xh).to(gpu)
            n = x.numel()
            ops = (Operand * 3)()
            ops[0].ptr, ops[0].scalar, ops[0].div, ops[0].mod, ops[0].dtype = x.data_ptr(), 0.0, 1, n, F64
            for j, p in enumerate((ald, bed), start=1):
                ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = p.data_ptr(), 0.0, 1, P, F64
            pp = ctypes.cast(ctypes.byref(ops, ctypes.sizeof(Operand)), ctypes.POINTER(Operand))
            out.fill_(float("nan"))
            _lib.check(L.m355_sumlogp(DIST["gamma"], FN[fn], n, ops, 2, pp, out.data_ptr(), ws.data_ptr(), ws.numel(), stream))
            _lib.check(L.m355_sumlogp_static(DIST["gamma"], FN[fn], n, ops, 2, pp, out.data_ptr() + 8, stream))
            a, b = (float(v) for v in out.cpu())
            terms = R.TABLE[f"gamma_{fn}"](xh, al, be).ravel()
            ref, scale = math.fsum(terms), float(np.sum(np.abs(terms)))
            assert abs(a - ref) <= 1e-12 * scale and abs(b - ref) <= 1e-12 * scale, (fn, rows, a, b, ref)


def test_sumlogp_static_rejects_oversize_and_bad_arguments(gpu):
    import torch

    from distributions_amd import _lib
    from distributions_amd._lib import DIST, FN

    L = _lib.lib()
    x = torch.ones(1024, dtype=torch.float64, device=gpu)
    out = torch.zeros(1, dtype=torch.float64, device=gpu)
    keep, px, pp = _ops(_lib, x, (0.0, 1.0))
    limit = int(L.m355_sumlogp_static_max_n())
    assert limit == 1 << 32
    rc = L.m355_sumlogp_static(DIST["normal"], FN["logpdf"], limit + 1, px, 2, pp, out.data_ptr(), None)
    assert rc == -4 and b"exceeds the maximum" in L.m355_last_error()
    rc = L.m355_sumlogp_static(DIST["normal"], FN["logpdf"], 1024, px, 2, pp, None, None)
    assert rc == -1
    rc = L.m355_sumlogp_static(DIST["normal"], FN["ppf"], 1024, px, 2, pp, out.data_ptr(), None)
    assert rc == -2
    torch.cuda.synchronize()
    del keep
