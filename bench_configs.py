#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs 2-4 (NOT the driver's headline line, which is
bench.py): one JSON line per (config, function) with kernel-only time from HIP events on the launch
stream.  Usage: python bench_configs.py [--config 2|3|4|all] [--n N] > profiles/rNN_configs.jsonl"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def timed(torch, fn, reps=5, warmup=1):
    for _ in range(warmup):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2]


def spot_check(name, fn, src, out, params, stride=97, cols=None):
    """After a timed region (never inside one): every `stride`-th element of `out` against the CPU oracle at the parity
    bar of the tests (tests/harness.py: 1e-12 max(1, |b|), the modelled window for studentt quantiles).  `cols` = P for
    row-periodic parameters of shape (P,).  Returns the worst scaled error; raises on a miss."""
    import numpy as np

    from oracle import restatement as R
    from tests import harness as H

    pts = src.view(-1)[::stride].cpu().numpy()
    got = out.view(-1)[::stride].cpu().numpy()
    if cols:
        col = (np.arange(pts.size, dtype=np.int64) * stride) % cols
        params = tuple(p.cpu().numpy()[col] if hasattr(p, "cpu") else p for p in params)
    with np.errstate(all="ignore"):
        exp = R.TABLE[f"{name}_{fn}"](pts, *params)
        ok = H.ppf_close(name, fn, params, got, exp) | H.ppf_unpinned(name, fn, pts)
        err = np.abs(got - exp) / np.maximum(1.0, np.abs(exp))
    if not np.all(ok):
        raise SystemExit(f"parity spot-check failed: {name}.{fn} {int((~ok).sum())} of {ok.size} points, e.g. x={pts[~ok][:3]} "
                         f"got={got[~ok][:3]} oracle={exp[~ok][:3]}")
    return float(np.nanmax(np.where(np.isfinite(err), err, 0.0))), int(ok.size)


def config2(torch, n):
    """Gamma/Beta/StudentT(/Normal) logpdf+logcdf+logsf+ppf on n fp64 points, scalar params, plus
    the period-1024 broadcast-parameter layout for logpdf."""
    from distributions_amd import raw
    from bench import make_inputs

    dev = torch.device("cuda", 0)
    xs = make_inputs(torch, n, dev)  # SURVEY.md 8(d) config-2 inputs, Philox seeds 1-4 (normal, gamma, beta, studentt)
    q = torch.empty(n, dtype=torch.float64, device=dev)
    raw.rvs_into("uniform", (0.0, 1.0), q, seed=5)
    q.clamp_(1e-12, 1 - 1e-12)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    ews = raw.elemwise_workspace(dev)  # the launch-time tables of logcdf / logsf / ppf live here (m355_elemwise_ws)
    params = {"normal": (0.3, 1.7), "gamma": (2.5, 1.3), "beta": (2.0, 5.0), "studentt": (5.0, 0.0, 2.0)}
    for name, p in params.items():
        for fn in ("logpdf", "logcdf", "logsf", "ppf"):
            src = q if fn == "ppf" else xs[name]
            ms = timed(torch, lambda: raw.elemwise_into(name, fn, src, p, out, ews), reps=3 if fn == "ppf" else 5)
            err, npts = spot_check(name, fn, src, out, p)
            print(json.dumps({"config": 2, "fn": f"{name}.{fn}", "n": n, "layout": "scalar params", "ms": ms,
                              "evals_per_s": n / (ms * 1e-3), "GB_per_s_algorithmic": 16.0 * n / (ms * 1e-3) / 1e9,
                              "parity_points": npts, "parity_max_err": err}),
                  flush=True)
    # float32 in, float32 out (8 B per element; the arithmetic stays fp64, tolerance 1e-5)
    out32 = torch.empty(n, dtype=torch.float32, device=dev)
    for name in ("normal", "gamma", "beta"):
        x32 = xs[name].to(torch.float32)
        ms = timed(torch, lambda: raw.elemwise_into(name, "logpdf", x32, params[name], out32), reps=5)
        print(json.dumps({"config": 2, "fn": f"{name}.logpdf", "n": n, "layout": "scalar params, float32 in / out", "ms": ms,
                          "evals_per_s": n / (ms * 1e-3), "GB_per_s_algorithmic": 8.0 * n / (ms * 1e-3) / 1e9}), flush=True)
        del x32
    del out32
    # broadcast params: x viewed as (n/1024, 1024), params of shape (1024,) staged in LDS
    from distributions_amd import gamma, normal

    rows = n // 1024
    xg = xs["gamma"][: rows * 1024].view(rows, 1024)
    a_row = torch.linspace(0.5, 9.0, 1024, dtype=torch.float64, device=dev)
    b_row = torch.linspace(0.3, 3.0, 1024, dtype=torch.float64, device=dev)
    for label, f in (("gamma.logpdf", lambda: gamma.logpdf(xg, a_row, b_row)),
                     ("normal.logpdf", lambda: normal.logpdf(xs["normal"][: rows * 1024].view(rows, 1024), a_row, b_row))):
        ms = timed(torch, f, reps=5)
        print(json.dumps({"config": 2, "fn": label, "n": rows * 1024, "layout": "params (1024,) row-periodic, LDS-staged "
                          "(time includes torch.empty of the output)", "ms": ms, "evals_per_s": rows * 1024 / (ms * 1e-3),
                          "GB_per_s_algorithmic": 16.0 * rows * 1024 / (ms * 1e-3) / 1e9}), flush=True)
    # the special-function functions with the same row-periodic parameters (config 2 as written): through the C-ABI
    # into a preallocated output, periodic operands
    periodic_special(torch, xs, q, out, rows)


def periodic_special(torch, xs, q, out, rows):
    import ctypes

    from distributions_amd import _lib
    from distributions_amd._lib import DIST, F64, FN, Operand

    L = _lib.lib()
    dev = out.device
    P = 1024
    n = rows * P
    lin = lambda lo, hi: torch.linspace(lo, hi, P, dtype=torch.float64, device=dev)
    cases = {"gamma": (lin(0.5, 9.0), lin(0.3, 3.0)), "beta": (lin(0.5, 9.0), lin(1.0, 10.0)),
             "studentt": (lin(2.0, 30.0), 0.0, 2.0)}
    stream = torch.cuda.current_stream(dev).cuda_stream

    def call(name, fn, src, params):
        ops = (Operand * (1 + len(params)))()
        ops[0].ptr, ops[0].div, ops[0].mod, ops[0].dtype = src.data_ptr(), 1, n, F64
        for j, p in enumerate(params, start=1):
            if isinstance(p, float):
                ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = None, p, 1, 1, F64
            else:
                ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = p.data_ptr(), 0.0, 1, P, F64
        at = lambda k: ctypes.cast(ctypes.byref(ops, k * ctypes.sizeof(Operand)), ctypes.POINTER(Operand))
        _lib.check(L.m355_elemwise_ws(DIST[name], FN[fn], F64, n, at(0), len(params), at(1), out.data_ptr(), ews.data_ptr(),
                                      ews.numel(), stream))

    sums = torch.zeros(1, dtype=torch.float64, device=dev)
    ws = torch.empty(int(L.m355_sumlogp_workspace_bytes()), dtype=torch.uint8, device=dev)
    ews = torch.empty(int(L.m355_elemwise_workspace_bytes()), dtype=torch.uint8, device=dev)  # the per-column tables

    def call_sum(name, params):
        ops = (Operand * (1 + len(params)))()
        ops[0].ptr, ops[0].div, ops[0].mod, ops[0].dtype = xs[name].data_ptr(), 1, n, F64
        for j, p in enumerate(params, start=1):
            if isinstance(p, float):
                ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = None, p, 1, 1, F64
            else:
                ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = p.data_ptr(), 0.0, 1, P, F64
        at = lambda k: ctypes.cast(ctypes.byref(ops, k * ctypes.sizeof(Operand)), ctypes.POINTER(Operand))
        _lib.check(L.m355_sumlogp(DIST[name], FN["logpdf"], n, at(0), len(params), at(1), sums.data_ptr(), ws.data_ptr(),
                                  ws.numel(), stream))

    for name, params in cases.items():
        ms = timed(torch, lambda: call_sum(name, params), reps=5)
        print(json.dumps({"config": 2, "fn": f"{name}.sum_logpdf", "n": n, "layout": "params (1024,) row-periodic", "ms": ms,
                          "evals_per_s": n / (ms * 1e-3), "GB_per_s_algorithmic": 8.0 * n / (ms * 1e-3) / 1e9}), flush=True)
    for name, params in cases.items():
        for fn in ("logpdf", "logcdf", "ppf"):
            src = q if fn == "ppf" else xs[name]
            ms = timed(torch, lambda: call(name, fn, src, params), reps=3)
            err, npts = spot_check(name, fn, src[:n], out[:n], params, cols=P)
            print(json.dumps({"config": 2, "fn": f"{name}.{fn}", "n": n, "layout": "params (1024,) row-periodic", "ms": ms,
                              "evals_per_s": n / (ms * 1e-3), "GB_per_s_algorithmic": 16.0 * n / (ms * 1e-3) / 1e9,
                              "parity_points": npts, "parity_max_err": err}), flush=True)


def config3(torch, n):
    """rvs: n draws each, output-only streams (8 B/draw)."""
    import ctypes

    from distributions_amd import _lib
    from distributions_amd._lib import DIST, F64, I64, Operand

    L = _lib.lib()
    dev = torch.device("cuda", 0)
    outf = torch.empty(n, dtype=torch.float64, device=dev)
    outi = torch.empty(n, dtype=torch.int64, device=dev)
    cases = [("normal", (0.0, 1.0)), ("gamma", (2.5, 1.0)), ("gamma", (0.5, 1.0)), ("beta", (2.0, 5.0)),
             ("studentt", (5.0, 0.0, 2.0)), ("poisson", (100.0,)), ("poisson", (4.0,)),
             ("binomial", (1000.0, 0.3)), ("binomial", (10.0, 0.3))]
    for name, p in cases:
        ops = (Operand * len(p))()
        for j, v in enumerate(p):
            ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = None, float(v), 1, 1, F64
        disc = name in ("poisson", "binomial")
        out = outi if disc else outf
        stream = torch.cuda.current_stream(dev).cuda_stream

        def go():
            _lib.check(L.m355_rvs(DIST[name], I64 if disc else F64, n, len(p), ops, 20260313, DIST[name], 0,
                                  out.data_ptr(), stream))

        ms = timed(torch, go, reps=3)
        print(json.dumps({"config": 3, "fn": f"{name}.rvs{p}", "n": n, "ms": ms, "draws_per_s": n / (ms * 1e-3),
                          "GB_per_s_algorithmic": 8.0 * n / (ms * 1e-3) / 1e9}), flush=True)


def config4(torch, n):
    """MvNormal logpdf: n observations, d = 64, fp64 (variant A: one shared covariance)."""
    import numpy as np

    from distributions_amd import mvnormal

    d = 64
    rng = np.random.default_rng(6)
    A = rng.normal(size=(d, d))
    cov = A @ A.T / d + np.eye(d)
    mu = np.random.default_rng(7).normal(size=d)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(8)
    Lc = torch.linalg.cholesky(torch.from_numpy(cov).to(dev))
    X = torch.from_numpy(mu).to(dev) + torch.randn(n, d, dtype=torch.float64, device=dev, generator=g) @ Lc.T
    mu_t, cov_t = torch.from_numpy(mu).to(dev), torch.from_numpy(cov).to(dev)
    ms = timed(torch, lambda: mvnormal.logpdf(X, mu_t, cov_t), reps=10, warmup=2)
    nbytes = 8.0 * (n * d + n + d * d + d)
    flops = n * (d * d + 3 * d) + d**3 / 3
    # parity after the timed region: every 97th observation against the oracle (scipy.linalg restatement of
    # mvnormal.py:17-37, 80-83) at 1e-12 max(1, |b|)
    from oracle import restatement as R

    got = mvnormal.logpdf(X, mu_t, cov_t)[::97].cpu().numpy()
    exp = R.mvnormal_logpdf(X[::97].cpu().numpy(), mu, cov)
    err_a = float(np.max(np.abs(got - exp) / np.maximum(1.0, np.abs(exp))))
    if not err_a <= 1e-12:
        raise SystemExit(f"parity spot-check failed: mvnormal.logpdf variant A, max scaled error {err_a}")
    print(json.dumps({"config": 4, "variant": "A (shared covariance)", "fn": "mvnormal.logpdf", "n": n, "d": d, "ms": ms,
                      "parity_points": int(got.size), "parity_max_err": err_a,
                      "obs_per_s": n / (ms * 1e-3), "GB_per_s_algorithmic": nbytes / (ms * 1e-3) / 1e9,
                      "frac_hbm_peak": nbytes / (ms * 1e-3) / 8e12, "GFLOP_per_s": flops / (ms * 1e-3) / 1e9,
                      "frac_fp64_mfma_peak": flops / (ms * 1e-3) / 78.6e12,
                      "note": "whole call: Cholesky + solve + epilogue + output allocation"}), flush=True)
    # the same with 10x the observations: the single-workgroup Cholesky (latency, ~constant) amortises
    n10 = 10 * n
    X10 = mu_t + torch.randn(n10, d, dtype=torch.float64, device=dev, generator=g) @ Lc.T
    ms = timed(torch, lambda: mvnormal.logpdf(X10, mu_t, cov_t), reps=10, warmup=2)
    nbytes = 8.0 * (n10 * d + n10 + d * d + d)
    flops = n10 * (d * d + 3 * d) + d**3 / 3
    print(json.dumps({"config": 4, "variant": "A (shared covariance)", "fn": "mvnormal.logpdf", "n": n10, "d": d, "ms": ms,
                      "obs_per_s": n10 / (ms * 1e-3), "GB_per_s_algorithmic": nbytes / (ms * 1e-3) / 1e9,
                      "frac_hbm_peak": nbytes / (ms * 1e-3) / 8e12, "GFLOP_per_s": flops / (ms * 1e-3) / 1e9,
                      "frac_fp64_mfma_peak": flops / (ms * 1e-3) / 78.6e12,
                      "note": "whole call: Cholesky + solve + epilogue + output allocation"}), flush=True)
    del X10
    # variant B: one covariance per observation (cov carries the batch axis, mvnormal.py:27): n distinct d x d matrices
    Ab = torch.randn(n, d, d, dtype=torch.float64, device=dev, generator=g)
    covb = torch.baddbmm(torch.eye(d, dtype=torch.float64, device=dev).expand(n, d, d), Ab, Ab.transpose(1, 2), alpha=1.0 / d)
    del Ab
    ms = timed(torch, lambda: mvnormal.logpdf(X, mu_t, covb), reps=5, warmup=1)
    sel = slice(0, n, 97 * 8)  # (the oracle factors one matrix per point in a Python loop)
    got = mvnormal.logpdf(X, mu_t, covb)[sel].cpu().numpy()
    exp = np.array([float(R.mvnormal_logpdf(xr[None, :], mu, cv)[0]) for xr, cv in zip(X[sel].cpu().numpy(), covb[sel].cpu().numpy())])
    err_b = float(np.max(np.abs(got - exp) / np.maximum(1.0, np.abs(exp))))
    if not err_b <= 1e-12:
        raise SystemExit(f"parity spot-check failed: mvnormal.logpdf variant B, max scaled error {err_b}")
    nbytes = 8.0 * n * (d * (d + 1) / 2 + 2 * d + 1)
    flops = n * (d**3 / 3 + d * d + 3 * d)
    print(json.dumps({"config": 4, "variant": "B (one covariance per observation)", "fn": "mvnormal.logpdf", "n": n, "d": d,
                      "ms": ms, "obs_per_s": n / (ms * 1e-3), "GB_per_s_algorithmic": nbytes / (ms * 1e-3) / 1e9,
                      "frac_hbm_peak": nbytes / (ms * 1e-3) / 8e12, "GFLOP_per_s": flops / (ms * 1e-3) / 1e9,
                      "frac_fp64_mfma_peak": flops / (ms * 1e-3) / 78.6e12, "parity_points": int(got.size), "parity_max_err": err_b,
                      "note": "whole call; algorithmic bytes count the lower triangle of each covariance (SURVEY.md 8(d)): "
                              "1.77 GB at n = 1e5; the full matrices are 3.28 GB"}), flush=True)


def config6(torch, n):
    """Closed-form / re-parameterised families: logpdf, logcdf and ppf on n fp64
    points drawn from each distribution's own rvs kernel, scalar parameters (16 B/element)."""
    from distributions_amd import _dispatch, raw
    from distributions_amd._lib import PARAMS

    cases = {"exponential": (0.27,), "lognormal": (2.0, 0.5), "halfnormal": (3.5,), "cauchy": (-5.0, 0.5), "halfcauchy": (3.5,),
             "laplace": (10.0, 2.0), "logistic": (2.5, 4.0), "gumbel": (2.5, 3.5), "weibull": (2.0, 5.0), "pareto": (2.5, 2.0),
             "uniform": (-5.0, 5.0), "triangular": (0.0, 0.1, 1.0), "kumaraswamy": (2.0, 3.0), "moyal": (-5.0, 0.5),
             "asymmetriclaplace": (-1.0, 2.0, 2.0), "loglogistic": (2.0, 3.0), "logitnormal": (-1.0, 1.0),
             "inversegamma": (5.0, 2.0), "chisquared": (5.0,), "scaledinversechisquared": (10.0, 2.0),
             "halfstudentt": (5.0, 2.0), "betascaled": (2.0, 5.0, -1.0, 3.0)}
    dev = torch.device("cuda", 0)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    ews = raw.elemwise_workspace(dev)
    g = torch.Generator(device=dev).manual_seed(9)
    q = torch.rand(n, dtype=torch.float64, device=dev, generator=g).clamp_(1e-12, 1 - 1e-12)
    for name, p in cases.items():
        assert len(p) == len(PARAMS[name])
        x = _dispatch.rvs(name, p, size=n, random_state=3, as_torch=True)
        row = {"config": 6, "dist": name, "n": n}
        for fn in ("logpdf", "logcdf", "ppf"):
            src = q if fn == "ppf" else x
            ms = timed(torch, lambda: raw.elemwise_into(name, fn, src, p, out, ews), reps=3)
            row[f"{fn}_ms"] = round(ms, 4)
            row[f"{fn}_GB_per_s"] = round(16.0 * n / (ms * 1e-3) / 1e9, 1)
        print(json.dumps(row), flush=True)
        del x


def config7(torch, n):
    """Gradient path: value + gradient of the summed log-likelihood in one pass, n fp64
    points, scalar parameters.  Algorithmic bytes: 8 B/element without dx (read x), 16 B/element with dx."""
    from distributions_amd import _dispatch, raw

    cases = {"normal": (0.3, 1.7), "gamma": (2.5, 1.3), "beta": (2.0, 5.0), "studentt": (5.0, 0.0, 2.0),
             "lognormal": (2.0, 0.5), "weibull": (2.0, 5.0), "cauchy": (-5.0, 0.5), "logistic": (2.5, 4.0),
             "inversegamma": (5.0, 2.0), "betascaled": (2.0, 5.0, -1.0, 3.0)}
    dev = torch.device("cuda", 0)
    dx = torch.empty(n, dtype=torch.float64, device=dev)
    sums = torch.zeros(8, dtype=torch.float64, device=dev)
    ws = raw.sumlogp_workspace(dev)
    for name, p in cases.items():
        x = _dispatch.rvs(name, p, size=n, random_state=3, as_torch=True)
        ms0 = timed(torch, lambda: raw.logpdf_grad_into(name, x, p, sums, None, ws), reps=5)
        ms1 = timed(torch, lambda: raw.logpdf_grad_into(name, x, p, sums, dx, ws), reps=5)
        ms2 = timed(torch, lambda: raw.sumlogp_into(name, "logpdf", x, p, sums, ws), reps=5)
        print(json.dumps({"config": 7, "dist": name, "n": n, "value_and_param_grads_ms": round(ms0, 4),
                          "GB_per_s": round(8.0 * n / (ms0 * 1e-3) / 1e9, 1), "with_dx_ms": round(ms1, 4),
                          "with_dx_GB_per_s": round(16.0 * n / (ms1 * 1e-3) / 1e9, 1),
                          "sum_logpdf_only_ms": round(ms2, 4)}), flush=True)
        del x


def config8(torch, n):
    """Solver / series families: logpdf, cdf and ppf, scalar parameters, points drawn from each
    distribution's own rvs kernel.  These are instruction-bound (series, quadrature, Newton iterations per element):
    the figure of merit is evaluations/s; sizes are reduced where one evaluation costs 1e4 .. 1e6 operations."""
    from distributions_amd import _dispatch, raw

    cases = {"wald": ((2.0, 10.0), 1.0, 1.0), "exgaussian": ((1.0, 3.0, 3.0), 1.0, 1.0),
             "truncatednormal": ((2.0, 1.5, -1.0, 5.0), 1.0, 1.0), "skew_studentt": ((8.0, 4.0, 0.0, 1.0), 1.0, 1.0),
             "skewnormal": ((0.5, 1.0, -2.0), 1.0, 1.0), "vonmises": ((0.5, 2.0), 1.0, 0.1),
             "rice": ((3.0, 2.0), 0.1, 0.01), "polyagamma": ((1.0, 0.0), 0.1, 0.01)}
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(9)
    for name, (p, f_cdf, f_ppf) in cases.items():
        x = _dispatch.rvs(name, p, size=n, random_state=3, as_torch=True)
        q = torch.rand(n, dtype=torch.float64, device=dev, generator=g).clamp_(1e-6, 1 - 1e-6)
        out = torch.empty(n, dtype=torch.float64, device=dev)
        row = {"config": 8, "dist": name}
        for fn, frac in (("logpdf", 1.0), ("cdf", f_cdf), ("ppf", f_ppf)):
            m = max(1024, int(n * frac))
            src = (q if fn == "ppf" else x)[:m]
            ms = timed(torch, lambda: raw.elemwise_into(name, fn, src, p, out[:m]), reps=3, warmup=1)
            row[f"{fn}_n"] = m
            row[f"{fn}_ms"] = round(ms, 4)
            row[f"{fn}_evals_per_s"] = float(f"{m / (ms * 1e-3):.4g}")
        print(json.dumps(row), flush=True)
        del x, q, out


def config9(torch, n):
    """Moment functions over a batch of parameter sets (m355_moment): closed forms (thread per set), medians through
    ppf, and the quadrature / support-sum stats (wavefront per set; `evals` = integrand evaluations, 1000 per
    trapezoid, helper.py:99-322)."""
    from distributions_amd import _dispatch

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(11)

    def u(lo, hi, m):
        return lo + (hi - lo) * torch.rand(m, dtype=torch.float64, device=dev, generator=g)

    cases = [("gamma", "entropy", n, lambda m: (u(0.5, 50, m), u(0.2, 5, m)), 1),
             ("gamma", "median", n // 10, lambda m: (u(0.5, 50, m), u(0.2, 5, m)), 1),
             ("weibull", "kurtosis", n, lambda m: (u(0.7, 9, m), u(0.5, 3, m)), 1),
             ("logitnormal", "kurtosis", n // 10, lambda m: (u(-2, 2, m), u(0.2, 2, m)), 3 * 70),
             ("binomial", "entropy", n // 10, lambda m: (torch.floor(u(5, 400, m)), u(0.05, 0.95, m)), 200),
             ("skew_studentt", "entropy", n // 100, lambda m: (u(2, 9, m), u(2, 9, m), u(-1, 1, m), u(0.5, 2, m)), 1000),
             ("rice", "kurtosis", n // 100, lambda m: (u(0.1, 6, m), u(0.5, 2, m)), 3000),
             ("polyagamma", "entropy", n // 1000, lambda m: (u(0.5, 4, m), u(-2, 2, m)), 1000)]
    for name, stat, m, make, evals in cases:
        m = max(64, m)
        params = make(m)
        ms = timed(torch, lambda: _dispatch.moment(name, stat, *params), reps=3, warmup=1)
        print(json.dumps({"config": 9, "dist": name, "stat": stat, "parameter_sets": m, "ms": round(ms, 4),
                          "sets_per_s": float(f"{m / (ms * 1e-3):.4g}"), "integrand_evals_per_s": float(f"{m * evals / (ms * 1e-3):.4g}")}), flush=True)


def config10(torch, n):
    """Discrete families: logpdf, cdf and ppf on int64 observations drawn from each family's
    own rvs kernel, scalar parameters.  logpdf is lgamma-bound, cdf runs gammaincc / betainc (or the reference's sum of
    pmf for betabinomial / hypergeometric), ppf the in-kernel find_ppf_discrete."""
    from distributions_amd import _dispatch, raw

    cases = {"poisson": (100.0,), "binomial": (1000.0, 0.3), "negativebinomial": (10.0, 0.1), "geometric": (0.1,),
             "betabinomial": (100.0, 20.0, 20.0), "hypergeometric": (50.0, 20.0, 10.0), "zi_poisson": (0.7, 10.0),
             "discreteweibull": (0.9, 3.0)}
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(10)
    for name, p in cases.items():
        x = _dispatch.rvs(name, p, size=n, random_state=5, as_torch=True)
        q = torch.rand(n, dtype=torch.float64, device=dev, generator=g).clamp_(1e-6, 1 - 1e-6)
        out = torch.empty(n, dtype=torch.float64, device=dev)
        row = {"config": 10, "dist": name, "n": n}
        for fn in ("logpdf", "cdf", "ppf"):
            src = q if fn == "ppf" else x
            ms = timed(torch, lambda: raw.elemwise_into(name, fn, src, p, out), reps=3, warmup=1)
            row[f"{fn}_ms"] = round(ms, 4)
            row[f"{fn}_evals_per_s"] = float(f"{n / (ms * 1e-3):.4g}")
        print(json.dumps(row), flush=True)
        del x, q, out
    # row-periodic parameters (x of shape (R, 1024), parameters (1024,)): per-column windows in the call's workspace
    # (k_dcolg, round 3); every 97th element against the oracle after the timed region
    P = 1024
    R = n // P
    ws = raw.elemwise_workspace(dev)
    lin = lambda a, b: torch.linspace(a, b, P, dtype=torch.float64, device=dev)
    pcases = {"poisson": (lin(5.0, 200.0),), "binomial": (torch.full((P,), 1000.0, dtype=torch.float64, device=dev), lin(0.05, 0.95))}
    for name, p in pcases.items():
        x = _dispatch.rvs(name, tuple(a.reshape(1, P) for a in p), size=(R, P), random_state=6, as_torch=True).reshape(-1).contiguous()
        q = torch.rand(R * P, dtype=torch.float64, device=dev, generator=g).clamp_(1e-6, 1 - 1e-6)
        out = torch.empty(R * P, dtype=torch.float64, device=dev)
        row = {"config": 10, "dist": name, "n": R * P, "layout": "params (1024,) row-periodic, int64 observations"}
        for fn in ("cdf", "logcdf", "ppf"):
            src = q if fn == "ppf" else x
            ms = timed(torch, lambda: raw.elemwise_into(name, fn, src, p, out, workspace=ws), reps=3, warmup=1)
            row[f"{fn}_ms"] = round(ms, 4)
            row[f"{fn}_evals_per_s"] = float(f"{R * P / (ms * 1e-3):.4g}")
            err, pts = spot_check(name, fn, src, out, p, cols=P)
            row[f"{fn}_parity_max_err"], row["parity_points"] = err, pts
        print(json.dumps(row), flush=True)
        del x, q, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="all")
    ap.add_argument("--n", type=int, default=0)
    args = ap.parse_args()
    import torch

    torch.cuda.set_device(0)
    if args.config in ("2", "all"):
        config2(torch, args.n or 100_000_000)
    if args.config in ("3", "all"):
        config3(torch, args.n or 1_000_000_000)
    if args.config in ("4", "all"):
        config4(torch, args.n or 100_000)
    if args.config in ("6", "all"):
        config6(torch, args.n or 100_000_000)
    if args.config in ("7", "all"):
        config7(torch, args.n or 100_000_000)
    if args.config in ("8", "all"):
        config8(torch, args.n or 10_000_000)
    if args.config in ("9", "all"):
        config9(torch, args.n or 10_000_000)
    if args.config in ("10", "all"):
        config10(torch, args.n or 100_000_000)


if __name__ == "__main__":
    main()
