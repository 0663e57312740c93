"""ctypes binding of libm355dist.so (C-ABI declared in include/m355_dist.h).

There is no CPU fallback: if the shared library is missing, or no GPU is visible, every call
raises.  PyTorch is used only to own device buffers and to name the current HIP stream.
"""

from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libm355dist.so")

F64, F32, I64 = 0, 1, 2
# distribution ids of include/m355_dist.h (enum m355_dist) and the reference's parameter names, in order
PARAMS = {
    "normal": ('mu', 'sigma'),
    "gamma": ('alpha', 'beta'),
    "beta": ('alpha', 'beta'),
    "studentt": ('nu', 'mu', 'sigma'),
    "poisson": ('mu',),
    "binomial": ('n', 'p'),
    "exponential": ('lam',),
    "lognormal": ('mu', 'sigma'),
    "halfnormal": ('sigma',),
    "cauchy": ('alpha', 'beta'),
    "halfcauchy": ('beta',),
    "laplace": ('mu', 'b'),
    "logistic": ('mu', 's'),
    "gumbel": ('mu', 'beta'),
    "weibull": ('alpha', 'beta'),
    "pareto": ('alpha', 'm'),
    "uniform": ('lower', 'upper'),
    "triangular": ('lower', 'c', 'upper'),
    "kumaraswamy": ('a', 'b'),
    "moyal": ('mu', 'sigma'),
    "asymmetriclaplace": ('mu', 'b', 'kappa'),
    "loglogistic": ('alpha', 'beta'),
    "logitnormal": ('mu', 'sigma'),
    "inversegamma": ('alpha', 'beta'),
    "chisquared": ('nu',),
    "scaledinversechisquared": ('nu', 'tau2'),
    "halfstudentt": ('nu', 'sigma'),
    "betascaled": ('alpha', 'beta', 'lower', 'upper'),
    "bernoulli": ('p',),
    "geometric": ('p',),
    "discreteuniform": ('lower', 'upper'),
    "discreteweibull": ('q', 'beta'),
    "negativebinomial": ('n', 'p'),
    "betabinomial": ('n', 'alpha', 'beta'),
    "hypergeometric": ('N', 'k', 'n'),
    "zi_poisson": ('psi', 'mu'),
    "zi_binomial": ('psi', 'n', 'p'),
    "zi_negativebinomial": ('psi', 'n', 'p'),
    "wald": ('mu', 'lam'),
    "exgaussian": ('mu', 'sigma', 'nu'),
    "truncatednormal": ('mu', 'sigma', 'lower', 'upper'),
    "skew_studentt": ('a', 'b', 'mu', 'sigma'),
    "skewnormal": ('mu', 'sigma', 'alpha'),
    "vonmises": ('mu', 'kappa'),
    "rice": ('nu', 'sigma'),
    "polyagamma": ('h', 'z'),
}
DIST = {name: i for i, name in enumerate(PARAMS)}
NPARAMS = {name: len(ps) for name, ps in PARAMS.items()}
DISCRETE = ("poisson", "binomial", "bernoulli", "geometric", "discreteuniform", "discreteweibull", "negativebinomial",
            "betabinomial", "hypergeometric", "zi_poisson", "zi_binomial", "zi_negativebinomial")  # rvs returns int64
# distributions with a gradient kernel (m355_grad_supported): the continuous families, poisson and binomial
GRAD = tuple(n for n in list(PARAMS)[:28] if n not in DISCRETE) + ("poisson", "binomial")
FN = {"logpdf": 0, "logcdf": 1, "logsf": 2, "ppf": 3, "pdf": 4, "cdf": 5, "sf": 6, "isf": 7}
# enum m355_stat of include/m355_dist.h: the moment functions every module of the reference defines
STAT = {"mean": 0, "mode": 1, "median": 2, "var": 3, "std": 4, "skewness": 5, "kurtosis": 6, "entropy": 7}


class Operand(ctypes.Structure):
    _fields_ = [
        ("ptr", ctypes.c_void_p),
        ("scalar", ctypes.c_double),
        ("div", ctypes.c_int64),
        ("mod", ctypes.c_int64),
        ("dtype", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class M355Error(RuntimeError):
    """A libm355dist.so entry point returned a non-zero status."""


_lib = None


def lib():
    """Load libm355dist.so (once).  Raises if it has not been built: run
    `python -c "import __graft_entry__ as g; g.build()"` or `make -C distributions_amd/csrc`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP library has not been built "
            "(make -C distributions_amd/csrc).  There is no CPU fallback."
        )
    # torch first: it brings its own copy of the HIP runtime, and the process must hold ONE runtime -- the library's
    # launches and torch's buffers / streams have to belong to the same one (loading the library before torch bound it
    # to /opt/rocm's copy: "no ROCm-capable device is detected" at the first launch on the GPU box)
    import torch  # noqa: F401

    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
    op = ctypes.POINTER(Operand)
    L.m355_version.restype = i32
    L.m355_last_error.restype = ctypes.c_char_p
    L.m355_device_count.restype = i32
    L.m355_elemwise.restype = i32
    L.m355_elemwise.argtypes = [i32, i32, i32, i64, op, i32, op, vp, vp]
    L.m355_elemwise_workspace_bytes.restype = sz
    L.m355_elemwise_workspace_bytes_for.restype = sz
    L.m355_elemwise_workspace_bytes_for.argtypes = [i64, i64]
    L.m355_elemwise_ws.restype = i32
    L.m355_elemwise_ws.argtypes = [i32, i32, i32, i64, op, i32, op, vp, vp, sz, vp]
    L.m355_sumlogp_workspace_bytes.restype = sz
    L.m355_sumlogp.restype = i32
    L.m355_sumlogp.argtypes = [i32, i32, i64, op, i32, op, vp, vp, sz, vp]
    L.m355_sumlogp_static_max_n.restype = i64
    L.m355_set_tab_min_n.restype = i64
    L.m355_set_tab_min_n.argtypes = [i64]
    L.m355_tab_min_n.restype = i64
    L.m355_tab_min_n.argtypes = []
    L.m355_sumlogp_static.restype = i32
    L.m355_sumlogp_static.argtypes = [i32, i32, i64, op, i32, op, vp, vp]
    L.m355_rvs.restype = i32
    L.m355_rvs.argtypes = [i32, i32, i64, i32, op, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, vp, vp]
    L.m355_grad_supported.restype = i32
    L.m355_moment.restype = i32
    L.m355_moment.argtypes = [i32, i32, i32, i64, i32, op, vp, vp]
    L.m355_grad_supported.argtypes = [i32]
    L.m355_logpdf_grad.restype = i32
    L.m355_logpdf_grad.argtypes = [i32, i32, i64, op, i32, op, vp, vp, ctypes.POINTER(vp), vp, sz, vp]
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise M355Error(f"libm355dist status {rc}: {lib().m355_last_error().decode()}")
