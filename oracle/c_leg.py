"""ctypes access to oracle/liboracle_c.so (the C leg of the CPU oracle; test/baseline only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = ctypes.POINTER(ctypes.c_double)


def load(build=True):
    path = os.path.join(_HERE, "liboracle_c.so")
    if not os.path.exists(path) and build:
        subprocess.check_call(["make", "-C", _HERE, "liboracle_c.so"])
    L = ctypes.CDLL(path)
    for name in ("oc_normal_logpdf", "oc_gamma_logpdf", "oc_beta_logpdf"):
        f = getattr(L, name)
        f.restype = None
        f.argtypes = [ctypes.c_long, _dp, ctypes.c_double, ctypes.c_double, _dp]
    L.oc_sum.restype = ctypes.c_double
    L.oc_sum.argtypes = [ctypes.c_long, _dp]
    return L


def logpdf(L, name, x, p0, p1, out=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    if out is None:
        out = np.empty_like(x)
    getattr(L, f"oc_{name}_logpdf")(x.size, x.ctypes.data_as(_dp), p0, p1, out.ctypes.data_as(_dp))
    return out
