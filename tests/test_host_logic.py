"""CPU tests: broadcasting/dtype planning of the host layer, the C-ABI library loads and exports
every symbol include/m355_dist.h declares (no compute calls without a GPU), and the product path
refuses to run without a GPU instead of falling back."""

import ctypes
import os
import re

import numpy as np
import pytest

import distributions_amd
from distributions_amd import _dispatch as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "m355_dist.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(m355_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = distributions_amd.lib()
    names = declared_symbols()
    assert "m355_elemwise" in names and "m355_sumlogp" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/m355_dist.h but not exported"
    assert lib.m355_version() == 100
    assert isinstance(lib.m355_last_error(), bytes)
    assert lib.m355_sumlogp_workspace_bytes() >= 8 * 4096


def test_argument_validation_without_gpu():
    """Validation happens before any HIP call, so it can be exercised on a CPU-only box."""
    lib = distributions_amd.lib()
    op = D.Operand(None, 0.0, 1, 1, 0, 0)
    params = (D.Operand * 2)(D.Operand(None, 0.0, 1, 1, 0, 0), D.Operand(None, 1.0, 1, 1, 0, 0))
    P = ctypes.POINTER(D.Operand)
    assert lib.m355_elemwise(99, 0, 0, 4, ctypes.byref(op), 2, ctypes.cast(params, P), None, None) == -1
    assert b"unknown distribution" in lib.m355_last_error()
    assert lib.m355_elemwise(4, 8, 0, 4, ctypes.byref(op), 1, ctypes.cast(params, P), None, None) == -2  # no such fn
    assert lib.m355_elemwise(0, 0, 0, 4, ctypes.byref(op), 3, ctypes.cast(params, P), None, None) == -1  # nparams
    assert lib.m355_elemwise(0, 0, 0, -1, ctypes.byref(op), 2, ctypes.cast(params, P), None, None) == -1
    assert lib.m355_elemwise(0, 0, 0, 0, ctypes.byref(op), 2, ctypes.cast(params, P), None, None) == 0  # empty


def test_plan_layouts_and_dtypes():
    f64, f32 = np.float64, np.float32
    p = D.plan(np.zeros((96, 1024)), [np.zeros(1024), 2.0])
    assert p.shape == (96, 1024) and not p.out_f32
    assert [(o.kind, o.div, o.mod) for o in p.operands] == [("full", 1, 98304), ("inner", 1, 1024), ("scalar", 1, 1)]
    p = D.plan(np.zeros((4, 8), f32), [np.zeros((4, 1), f32), 2.0])
    assert p.out_f32 and (p.operands[1].kind, p.operands[1].div, p.operands[1].mod) == ("outer", 8, 4)
    p = D.plan(np.zeros((5, 7, 33)), [0.1, np.zeros((1, 7, 1))])
    assert (p.operands[2].kind, p.operands[2].div, p.operands[2].mod) == ("middle", 33, 7)
    p = D.plan(np.zeros((4, 1, 8)), [np.zeros((3, 1)), 1.0])
    assert p.shape == (4, 3, 8) and p.operands[0].kind == "expand"
    assert D.plan(np.zeros(3, f32), [np.zeros(3, f64), 1.0]).out_f32 is False  # mixed -> f64
    assert D.plan(np.zeros(3, np.int64), [np.float32(1.0)]).out_f32 is False  # int64 x -> f64
    assert D.plan(2.0, [0.0, 1.0]).shape == ()  # 0-d in, 0-d out
    assert D.plan(np.zeros((0, 3)), [0.0, 1.0]).n == 0
    with pytest.raises(ValueError):
        D.plan(np.zeros((3, 4)), [np.zeros(5), 1.0])


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from distributions_amd import normal

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        normal.logpdf(np.zeros(4), 0.0, 1.0)
    with pytest.raises(TypeError):
        normal.logpdf(np.zeros(4), 0.0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "distributions_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the CPU oracle", ""), f


def test_every_module_imports_with_the_reference_signatures():
    """Each distribution module defines the reference's functions with its parameter names and order
    (first argument x; q for ppf / isf, except discreteweibull whose own parameter is called q:
    discreteweibull.py:53,65 write isf(x, q, beta) and ppf(p, q, beta))."""
    import importlib
    import inspect

    from distributions_amd._lib import FN, PARAMS

    for dist, names in PARAMS.items():
        m = importlib.import_module(f"distributions_amd.{dist}")
        for fn in FN:
            sig = list(inspect.signature(getattr(m, fn)).parameters)
            assert sig[1:] == list(names), (dist, fn, sig)
            assert len(set(sig)) == len(sig)
        assert list(inspect.signature(m.rvs).parameters) == list(names) + ["size", "random_state"]
        assert list(inspect.signature(m.sum_logpdf).parameters) == ["x"] + list(names)
        for st in ("mean", "mode", "median", "var", "std", "skewness", "kurtosis", "entropy"):  # <dist>.py moment block
            assert list(inspect.signature(getattr(m, st)).parameters) == list(names), (dist, st)
