"""Builds the reference's module-level function surface (pdf, logpdf, cdf, logcdf, sf, logsf, ppf, isf,
rvs + the fused sum_logpdf) for one distribution of the elementwise path, with the reference's
parameter names and order (``_lib.PARAMS``), each function one call into the C-ABI."""

from __future__ import annotations

from ._dispatch import elemwise as _elemwise
from ._dispatch import logpdf_grad as _logpdf_grad
from ._dispatch import moment as _moment
from ._dispatch import rvs as _rvs
from ._dispatch import sumlogp as _sumlogp
from ._lib import FN, GRAD, PARAMS, STAT


def export(ns: dict, dist: str, cites: dict | None = None) -> None:
    """Define the functions of `dist` in the module namespace `ns`.  `cites` maps a function name to
    the reference lines it replaces (goes into the docstring)."""
    names = PARAMS[dist]
    sig = ", ".join(names)
    cites = cites or {}
    for fn in FN:
        first = "q" if fn in ("ppf", "isf") else "x"
        if first in names:  # discreteweibull has a parameter q: the reference writes ppf(p, q, beta), isf(x, q, beta)
            first = "p" if fn == "ppf" else "x"
        src = (f"def {fn}({first}, {sig}):\n"
               f"    return _elemwise({dist!r}, {fn!r}, {first}, {sig})\n")
        loc: dict = {}
        exec(src, {"_elemwise": _elemwise}, loc)  # noqa: S102 - fixed template, gives real signatures
        f = loc[fn]
        f.__doc__ = f"{dist}.py:{cites.get(fn, '?')} of the reference, evaluated by the HIP elementwise kernels."
        f.__module__ = ns.get("__name__", __name__)
        ns[fn] = f
    loc = {}
    exec(f"def sum_logpdf(x, {sig}):\n    return _sumlogp({dist!r}, 'logpdf', x, {sig})\n", {"_sumlogp": _sumlogp}, loc)  # noqa: S102
    loc["sum_logpdf"].__doc__ = "sum(logpdf(x, ...)) fused in one pass (the reference spells it pt.sum(logpdf(...)))."
    ns["sum_logpdf"] = loc["sum_logpdf"]
    for fn in ("logcdf", "logsf"):  # censored likelihoods: sum(logcdf(...)) / sum(logsf(...)) without the elementwise result
        loc = {}
        exec(f"def sum_{fn}(x, {sig}):\n    return _sumlogp({dist!r}, {fn!r}, x, {sig})\n", {"_sumlogp": _sumlogp}, loc)  # noqa: S102
        loc[f"sum_{fn}"].__doc__ = (f"sum({fn}(x, ...)) fused in one pass (the reference spells it pt.sum({fn}(...))); for gamma, beta, "
                                    "studentt and their relatives with scalar parameters from the launch-time table (k_tab).")
        ns[f"sum_{fn}"] = loc[f"sum_{fn}"]
    loc = {}
    exec(f"def rvs({sig}, size=None, random_state=None):\n"
         f"    return _rvs({dist!r}, ({sig},), size=size, random_state=random_state)\n", {"_rvs": _rvs}, loc)  # noqa: S102
    loc["rvs"].__doc__ = (f"{dist}.py:{cites.get('rvs', '?')} of the reference; Philox counter-based draws, statistically "
                          "equivalent to the reference sampler (SURVEY.md quirk Q3).")
    ns["rvs"] = loc["rvs"]
    export_moments(ns, dist)
    if dist in GRAD:
        export_grad(ns, dist)


def export_moments(ns: dict, dist: str) -> None:
    """mean / mode / median / var / std / skewness / kurtosis / entropy of `dist` with the reference's signature
    (the parameters only), evaluated per parameter set by m355_moment (csrc/m355_moments.h)."""
    sig = ", ".join(PARAMS[dist])
    for st in STAT:
        loc: dict = {}
        exec(f"def {st}({sig}):\n    return _moment({dist!r}, {st!r}, {sig})\n", {"_moment": _moment}, loc)  # noqa: S102
        f = loc[st]
        f.__doc__ = f"{dist}.{st} of the reference ({dist}.py, moment block), one value per parameter set."
        f.__module__ = ns.get("__name__", __name__)
        ns[st] = f


def export_grad(ns: dict, dist: str) -> None:
    """sum_logpdf_grad / dlogpdf of `dist` (gradient path; the reference gets these from
    pytensor.grad of its logpdf graph)."""
    sig = ", ".join(PARAMS[dist])
    loc: dict = {}
    exec(f"def sum_logpdf_grad(x, {sig}, x_grad=True):\n    return _g({dist!r}, x, {sig}, x_grad=x_grad)\n"
         f"def dlogpdf(x, {sig}):\n    return _g({dist!r}, x, {sig}, elementwise=True)[1]\n",
         {"_g": _logpdf_grad}, loc)  # noqa: S102
    loc["sum_logpdf_grad"].__doc__ = ("(sum(logpdf(x, ...)), [gradient w.r.t. x, w.r.t. each parameter]) in one pass; "
                                      "every gradient has the shape of its argument (x_grad=False: x is data, its slot is None).")
    loc["dlogpdf"].__doc__ = "[d logpdf/dx, d logpdf/dparam_0, ...] per element of the broadcast shape."
    ns["sum_logpdf_grad"] = loc["sum_logpdf_grad"]
    ns["dlogpdf"] = loc["dlogpdf"]
