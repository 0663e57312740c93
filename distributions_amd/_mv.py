"""Host side of the multivariate log-densities: the shape logic of quaddist_chol (mvnormal.py:17-37
of the reference: 0-d value raises, 1-d value is one observation and gives a 0-d result, leading
batch axes of value / mu / cov broadcast Blockwise-style), then one call into the C-ABI."""

from __future__ import annotations

import ctypes
from math import prod

import numpy as np

from . import _lib
from ._dispatch import _device_of, _is_torch, _torch, classify

_vp, _i32, _i64, _dbl, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_size_t
_bound = False


def _bind():
    global _bound
    L = _lib.lib()
    if not _bound:
        L.m355_mvn_workspace_bytes.restype = _sz
        L.m355_mvn_workspace_bytes.argtypes = [_i32, _i64]
        L.m355_mvn_logpdf.restype = _i32
        L.m355_mvn_logpdf.argtypes = [_i32, _i64, _i32, _vp, _vp, _i64, _vp, _i64, _i64, _i64, _dbl, _vp, _vp, _sz, _vp]
        L.m355_matrixnormal_workspace_bytes.restype = _sz
        L.m355_matrixnormal_workspace_bytes.argtypes = [_i64, _i32, _i32, _i64, _i64]
        L.m355_matrixnormal_logpdf.restype = _i32
        L.m355_matrixnormal_logpdf.argtypes = [_i64, _i32, _i32, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _sz, _vp]
        _bound = True
    return L


def _t(a, torch, device):
    if _is_torch(a):
        return a.to(device=device, dtype=torch.float64)
    a = np.asarray(a, dtype=np.float64)  # np.ascontiguousarray would promote 0-d to 1-d
    return torch.from_numpy(np.array(a, order="C", copy=True)).to(device)


def mv_logpdf(kind: int, x, mu, cov, nu: float = 0.0):
    torch = _torch()
    L = _bind()
    args = (x, mu, cov)
    device = _device_of(args, torch)
    want_torch = any(_is_torch(a) for a in args)
    x, mu, cov = (_t(a, torch, device) for a in args)
    if x.ndim == 0:
        raise ValueError("Value can't be a scalar")
    onedim = x.ndim == 1
    if onedim:
        x = x[None, :]
    d = x.shape[-1]
    if cov.ndim < 2 or cov.shape[-1] != d or cov.shape[-2] != d or mu.shape[-1:] not in ((d,), ()):
        raise ValueError(f"shape mismatch: value (..., {d}), mu {tuple(mu.shape)}, cov {tuple(cov.shape)}")
    cov_batch = tuple(cov.shape[:-2])
    mu_batch = tuple(mu.shape[:-1]) if mu.ndim >= 1 else ()
    batch = tuple(np.broadcast_shapes(tuple(x.shape[:-1]), mu_batch, cov_batch))
    n = prod(batch)
    with torch.cuda.device(device):
        xe = x.expand(batch + (d,)).contiguous()
        if mu.ndim <= 1:
            mu_c, mu_stride = mu.expand(d).contiguous(), 0
        else:
            mu_c, mu_stride = mu.expand(batch + (d,)).contiguous(), d
        ncov = prod(cov_batch)
        if ncov == 1:
            cov_c, cdiv, cmod = cov.reshape(1, d, d).contiguous(), 1, 1
        else:
            pl = classify(cov_batch, batch)
            if pl.kind == "expand":
                cov_c = cov.expand(batch + (d, d)).contiguous().reshape(n, d, d)
                ncov, cdiv, cmod = n, 1, n
            else:
                cov_c, cdiv, cmod = cov.reshape(ncov, d, d).contiguous(), pl.div, pl.mod
        out = torch.empty(n, dtype=torch.float64, device=device)
        quirk = onedim and batch != (1,) and n > 0
        at_mean = torch.empty(n, dtype=torch.float64, device=device) if quirk else None
        if n:
            ws_bytes = int(L.m355_mvn_workspace_bytes(d, ncov))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(L.m355_mvn_logpdf(kind, n, d, xe.data_ptr(), mu_c.data_ptr(), mu_stride, cov_c.data_ptr(), ncov,
                                         cdiv, cmod, float(nu), out.data_ptr(), ws.data_ptr(), ws_bytes, stream))
            if quirk:
                # logpdf at delta = 0 (value = mu): the part of the result that does not depend on quaddist
                mu_e = mu.expand(batch + (d,)).contiguous()
                _lib.check(L.m355_mvn_logpdf(kind, n, d, mu_e.data_ptr(), mu_e.data_ptr(), d, cov_c.data_ptr(), ncov,
                                             cdiv, cmod, float(nu), at_mean.data_ptr(), ws.data_ptr(), ws_bytes, stream))
    out = out.reshape(batch)
    if quirk:
        # mvnormal.py:34-35 returns quaddist[0] for a 1-d value: with batched mu / cov that is the quadratic form of
        # the FIRST entry along the leading batch axis, which then broadcasts against every entry's logdet (and nu
        # terms).  Reproduced as the reference computes it: result_b = logpdf_b(value = mu_b) + the quaddist-dependent
        # part of entry 0 (for both kinds that part is logpdf - logpdf(value = mu)).
        at_mean = at_mean.reshape(batch)
        out = at_mean + (out[0] - at_mean[0])
        # shape of the reference's result: logdet (cov's batch axes) against quaddist[0] (the batch without its
        # leading axis); along the other axes `out` is constant
        tshape = tuple(np.broadcast_shapes(cov_batch, batch[1:]))
        lead = len(batch) - len(tshape)
        out = out[(0,) * lead]
        for ax, (td, bd) in enumerate(zip(tshape, batch[lead:])):
            if td == 1 and bd != 1:
                out = out.narrow(ax, 0, 1)
    if onedim and batch == (1,):
        out = out[0]  # mvnormal.py:34-35: a 1-d value is one observation and gives a 0-d result
    return out if want_torch else out.cpu().numpy()


def matrixnormal_logpdf(X, mu, rowcov, colcov):
    torch = _torch()
    L = _bind()
    args = (X, mu, rowcov, colcov)
    device = _device_of(args, torch)
    want_torch = any(_is_torch(a) for a in args)
    X, mu, rowcov, colcov = (_t(a, torch, device) for a in args)
    m, n = mu.shape[-2], mu.shape[-1]
    if X.shape[-2:] != (m, n) or rowcov.shape[-2:] != (m, m) or colcov.shape[-2:] != (n, n):
        raise ValueError("matrixnormal: X/mu must be (..., m, n), rowcov (..., m, m), colcov (..., n, n)")
    batch = tuple(np.broadcast_shapes(tuple(X.shape[:-2]), tuple(mu.shape[:-2]), tuple(rowcov.shape[:-2]),
                                      tuple(colcov.shape[:-2])))
    nb = prod(batch)
    with torch.cuda.device(device):
        Xe = X.expand(batch + (m, n)).contiguous()
        mu_batched = int(mu.ndim > 2 and prod(mu.shape[:-2]) > 1)
        mu_c = mu.expand(batch + (m, n)).contiguous() if mu_batched else mu.reshape(m, n).contiguous()

        def covs(c, k):
            if prod(c.shape[:-2]) == 1:
                return c.reshape(1, k, k).contiguous(), 1
            return c.expand(batch + (k, k)).contiguous().reshape(nb, k, k), nb

        U, nrow = covs(rowcov, m)
        V, ncol = covs(colcov, n)
        out = torch.empty(nb, dtype=torch.float64, device=device)
        if nb:
            ws_bytes = int(L.m355_matrixnormal_workspace_bytes(nb, m, n, nrow, ncol))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(L.m355_matrixnormal_logpdf(nb, m, n, Xe.data_ptr(), mu_c.data_ptr(), mu_batched, U.data_ptr(),
                                                  nrow, V.data_ptr(), ncol, out.data_ptr(), ws.data_ptr(), ws_bytes,
                                                  stream))
    out = out.reshape(batch)
    return out if want_torch else out.cpu().numpy()
