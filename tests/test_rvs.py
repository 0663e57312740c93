"""rvs: Philox4x32-10 known-answer vectors and KS / chi-square equivalence with the reference
sampler's distributions (BASELINE.json: "rvs must pass KS/chi-square equivalence against the
reference sampler"; the reference sampler is numpy.random.Generator).  The CPU tests
run the host build of the sampler code (tests/hostmath); the `gpu` tests run the HIP kernels through
the module API and also compare them draw-by-draw with the host build."""

import ctypes

import numpy as np
import pytest
from scipy import stats

from oracle import restatement as R
from tests.test_hostmath import hostlib  # noqa: F401  (fixture)

_dp = ctypes.POINTER(ctypes.c_double)
_u32p = ctypes.POINTER(ctypes.c_uint32)
from distributions_amd._lib import DIST  # ids of include/m355_dist.h (pure Python, no library load)

class _folded:
    """|X| for a scipy distribution symmetric about 0 (the cdf the KS tests read)."""

    def __init__(self, d):
        self.d = d

    def cdf(self, x):
        x = np.asarray(x, dtype=float)
        return np.where(x < 0, 0.0, 2.0 * self.d.cdf(x) - 1.0)


CONT = [
    ("normal", (0.3, 1.7), stats.norm(0.3, 1.7)),
    ("gamma", (2.5, 1.3), stats.gamma(2.5, scale=1 / 1.3)),   # Marsaglia-Tsang
    ("gamma", (0.5, 1.0), stats.gamma(0.5)),                  # alpha < 1 boost
    ("gamma", (50.0, 2.0), stats.gamma(50.0, scale=0.5)),
    ("beta", (2.0, 5.0), stats.beta(2, 5)),
    ("beta", (0.5, 0.7), stats.beta(0.5, 0.7)),
    ("beta", (1.01, 1.02), stats.beta(1.01, 1.02)),           # Cheng's BB at its least efficient corner
    ("beta", (50.0, 3.0), stats.beta(50.0, 3.0)),             # BB, shapes swapped
    ("beta", (300.0, 700.0), stats.beta(300.0, 700.0)),
    ("beta", (1.0, 3.0), stats.beta(1.0, 3.0)),               # a shape of exactly 1: the gamma ratio
    ("studentt", (5.0, 0.0, 2.0), stats.t(5, 0, 2)),
]
DISC = [  # (comments: the per-lane sampler's branch; with scalar parameters the library inverts the tabulated cdf instead)
    ("poisson", (100.0,), stats.poisson(100)),      # PTRS
    ("poisson", (4.0,), stats.poisson(4)),          # sequential inversion
    ("poisson", (10.0,), stats.poisson(10)),        # PTRS at its lower edge
    ("binomial", (1000, 0.3), stats.binom(1000, 0.3)),  # BTRS
    ("binomial", (10, 0.3), stats.binom(10, 0.3)),      # inversion
    ("binomial", (100, 0.9), stats.binom(100, 0.9)),    # p > 1/2 flip
    ("binomial", (40, 0.25), stats.binom(40, 0.25)),    # BTRS at n p = 10
]
DISC_BIG = [  # windows beyond 2048 integers: the rejection samplers also with scalar parameters (k_rvs_compact)
    ("poisson", (20000.0,), stats.poisson(20000)),
    ("binomial", (200000, 0.3), stats.binom(200000, 0.3)),
    ("binomial", (500000, 0.9), stats.binom(500000, 0.9)),
]


def host_rvs(lib, name, n, params, seed=20260313, stream=None, offset=0):
    lib.hm_rvs.argtypes = [ctypes.c_int, ctypes.c_long, _dp, ctypes.c_ulonglong, ctypes.c_uint,
                           ctypes.c_ulonglong, _dp]
    p = np.array(list(params) + [0, 0, 0, 0], dtype=np.float64)[:4].copy()
    out = np.empty(n)
    stream = DIST[name] if stream is None else stream
    assert lib.hm_rvs(DIST[name], n, p.ctypes.data_as(_dp), seed, stream, offset, out.ctypes.data_as(_dp)) == 0
    return out


INV_DISTS = ("poisson", "binomial", "negativebinomial", "betabinomial", "hypergeometric", "zi_poisson", "zi_binomial",
             "zi_negativebinomial")  # scalar parameters: inversion on the tabulated cdf (k_rvs_inv) where the window fits


def host_rvs_inv(lib, name, n, params, seed=20260313, stream=None, offset=0):
    """Host build of k_rvs_inv's body; None where the family's window does not fit (the library then samples per lane)."""
    lib.hm_rvs_inv.argtypes = [ctypes.c_int, ctypes.c_long, _dp, ctypes.c_ulonglong, ctypes.c_uint,
                               ctypes.c_ulonglong, _dp]
    p = np.array(list(params) + [0, 0, 0, 0], dtype=np.float64)[:4].copy()
    out = np.empty(n)
    stream = DIST[name] if stream is None else stream
    rc = lib.hm_rvs_inv(DIST[name], n, p.ctypes.data_as(_dp), seed, stream, offset, out.ctypes.data_as(_dp))
    assert rc in (0, -2)
    return out if rc == 0 else None


def host_rvs_scalar(lib, name, n, params, **kw):
    """What the library draws for SCALAR parameters: the inversion sampler for INV_DISTS, else the per-lane sampler."""
    x = host_rvs_inv(lib, name, n, params, **kw) if name in INV_DISTS else None
    return host_rvs(lib, name, n, params, **kw) if x is None else x


def chi2_pvalue(x, d):
    x = np.asarray(x, dtype=np.int64)
    lo, hi = int(x.min()), int(x.max())
    ks = np.arange(lo, hi + 1)
    obs = np.bincount(x - lo, minlength=hi - lo + 1).astype(float)
    exp = d.pmf(ks) * x.size
    keep = exp >= 5  # bins of expected count >= 5, the rest pooled into the two tails
    first, last = ks[keep][0], ks[keep][-1]
    o = np.concatenate([[obs[ks < first].sum()], obs[keep], [obs[ks > last].sum()]])
    e = np.concatenate([[x.size * d.cdf(first - 1)], exp[keep], [x.size * d.sf(last)]])
    m = e > 0
    stat = ((o[m] - e[m]) ** 2 / e[m]).sum()
    return stats.chi2.sf(stat, m.sum() - 1)


def moments_ok(x, d, nsig=5.0):
    n = x.size
    m, v = d.mean(), d.var()
    if not np.isfinite(v):
        return True
    ok = abs(x.mean() - m) <= nsig * np.sqrt(v / n)
    k4 = d.stats(moments="k")
    if np.isfinite(k4):
        ok &= abs(x.var() - v) <= nsig * v * np.sqrt((k4 + 2.0) / n)
    return bool(ok)


def test_philox4x32_10_known_answers(hostlib):
    """Random123 kat_vectors for philox4x32-10."""
    def run(ctr, key):
        c, k, o = np.array(ctr, np.uint32), np.array(key, np.uint32), np.zeros(4, np.uint32)
        hostlib.hm_philox(c.ctypes.data_as(_u32p), k.ctypes.data_as(_u32p), o.ctypes.data_as(_u32p))
        return [int(v) for v in o]

    assert run([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert run([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert run([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


@pytest.mark.parametrize("name, params, d", CONT, ids=[f"{c[0]}-{c[1]}" for c in CONT])
def test_host_sampler_continuous_ks(hostlib, name, params, d):
    x = host_rvs(hostlib, name, 400_000, params)
    assert stats.kstest(x, d.cdf).pvalue > 1e-3
    assert moments_ok(x, d)
    # two-sample KS against the reference sampler (numpy Generator)
    ref = R.rvs(name, params, 400_000, seed=7)
    assert stats.ks_2samp(x, ref).pvalue > 1e-3


@pytest.mark.parametrize("name, params, d", DISC + DISC_BIG, ids=[f"{c[0]}-{c[1]}" for c in DISC + DISC_BIG])
def test_host_sampler_discrete_chi2(hostlib, name, params, d):
    x = host_rvs(hostlib, name, 400_000, params)
    assert np.all(x == np.floor(x)) and x.min() >= 0
    assert chi2_pvalue(x, d) > 1e-3
    assert moments_ok(x, d)


@pytest.mark.parametrize("name, params, d", DISC + [("poisson", (0.05,), stats.poisson(0.05)), ("poisson", (5000.0,), stats.poisson(5000)),
                                                    ("binomial", (1, 0.3), stats.binom(1, 0.3)), ("binomial", (2000, 0.999), stats.binom(2000, 0.999))],
                         ids=lambda v: str(v) if isinstance(v, (str, tuple)) else "")
def test_host_inversion_sampler(hostlib, name, params, d):
    """The sampler for scalar parameters (body of k_rvs_inv): chi-square and moments, draw == ppf(u) by construction
    (smallest k with cdf(k) >= u), any split of the stream concatenates to the same draws."""
    x = host_rvs_inv(hostlib, name, 1_000_000, params)
    assert x is not None and np.all(x == np.floor(x)) and x.min() >= 0
    assert chi2_pvalue(x, d) > 1e-3
    assert moments_ok(x, d)
    parts = np.concatenate([host_rvs_inv(hostlib, name, m, params, offset=o) for o, m in ((0, 7), (7, 1000), (1007, 4993))])
    assert np.array_equal(x[:6000], parts)


def test_host_inversion_sampler_window(hostlib):
    assert host_rvs_inv(hostlib, "poisson", 10, (20000.0,)) is None  # 23 sd + 52 > 2048 integers
    assert host_rvs_inv(hostlib, "binomial", 10, (200000, 0.3)) is None


class OracleDiscrete:
    """pmf / cdf / sf / moments of a discrete family taken from the oracle (for the families SciPy
    does not have: discreteweibull and the zero-inflated ones)."""

    def __init__(self, name, params, hi=2000):
        self.name, self.params = name, params
        self.ks = np.arange(0, hi, dtype=np.float64)
        self.p = R.TABLE[f"{name}_pdf"](self.ks, *params)

    def pmf(self, k):
        return R.TABLE[f"{self.name}_pdf"](np.asarray(k, dtype=np.float64), *self.params)

    def cdf(self, k):
        return R.TABLE[f"{self.name}_cdf"](np.asarray(k, dtype=np.float64), *self.params)

    def sf(self, k):
        return R.TABLE[f"{self.name}_sf"](np.asarray(k, dtype=np.float64), *self.params)

    def mean(self):
        return float((self.ks * self.p).sum())

    def var(self):
        return float((self.ks ** 2 * self.p).sum() - self.mean() ** 2)

    def stats(self, moments="k"):
        m, v = self.mean(), self.var()
        return float((((self.ks - m) ** 4) * self.p).sum() / v ** 2 - 3.0)


# rvs of the discrete families; the SciPy objects are the ones the reference's own
# tests compare with (tests/test_bernoulli.py ... test_hypergeometric.py of the reference)
DISC_FAMILIES = [
    ("bernoulli", (0.3,), lambda: stats.bernoulli(0.3)),
    ("bernoulli", (0.999,), lambda: stats.bernoulli(0.999)),
    ("geometric", (0.1,), lambda: stats.geom(0.1)),
    ("geometric", (0.9,), lambda: stats.geom(0.9)),
    ("discreteuniform", (-2.0, 5.0), lambda: stats.randint(-2, 6)),
    ("negativebinomial", (2.1, 0.375), lambda: stats.nbinom(2.1, 0.375)),
    ("negativebinomial", (20.0, 0.05), lambda: stats.nbinom(20.0, 0.05)),   # Poisson leg on PTRS
    ("negativebinomial", (0.5, 0.9), lambda: stats.nbinom(0.5, 0.9)),       # gamma shape < 1
    ("betabinomial", (10.0, 2.0, 3.0), lambda: stats.betabinom(10, 2.0, 3.0)),
    ("betabinomial", (100.0, 20.0, 20.0), lambda: stats.betabinom(100, 20.0, 20.0)),
    ("betabinomial", (20.0, 0.5, 0.5), lambda: stats.betabinom(20, 0.5, 0.5)),
    ("hypergeometric", (20.0, 7.0, 12.0), lambda: stats.hypergeom(M=20, n=7, N=12)),
    ("hypergeometric", (50.0, 20.0, 10.0), lambda: stats.hypergeom(M=50, n=20, N=10)),
    ("discreteweibull", (0.8, 2.0), lambda: OracleDiscrete("discreteweibull", (0.8, 2.0))),
    ("discreteweibull", (0.9, 3.0), lambda: OracleDiscrete("discreteweibull", (0.9, 3.0))),
    ("zi_poisson", (0.7, 2.0), lambda: OracleDiscrete("zi_poisson", (0.7, 2.0))),
    ("zi_poisson", (0.3, 10.0), lambda: OracleDiscrete("zi_poisson", (0.3, 10.0))),
    ("zi_binomial", (0.5, 20.0, 0.5), lambda: OracleDiscrete("zi_binomial", (0.5, 20.0, 0.5))),
    ("zi_negativebinomial", (0.7, 5.0, 0.4), lambda: OracleDiscrete("zi_negativebinomial", (0.7, 5.0, 0.4))),
]
_DISC_IDS = [f"{c[0]}-{c[1]}" for c in DISC_FAMILIES]


@pytest.mark.parametrize("name, params, make", DISC_FAMILIES, ids=_DISC_IDS)
def test_host_sampler_discrete_families_chi2(hostlib, name, params, make):
    d = make()
    x = host_rvs(hostlib, name, 400_000, params)
    assert np.all(x == np.floor(x))
    if chi2_pvalue(x, d) <= 1e-3:  # one stream in a thousand lands here by chance: a second, larger one must not
        x = host_rvs(hostlib, name, 4_000_000, params, seed=11)
        assert chi2_pvalue(x, d) > 1e-2
    assert moments_ok(x, d)
    if name in INV_DISTS:  # the sampler for scalar parameters
        xi = host_rvs_inv(hostlib, name, 1_000_000, params)
        assert xi is not None and np.all(xi == np.floor(xi))
        assert chi2_pvalue(xi, d) > 1e-3
        assert moments_ok(xi, d)
        parts = np.concatenate([host_rvs_inv(hostlib, name, m, params, offset=o) for o, m in ((0, 7), (7, 1000), (1007, 4993))])
        assert np.array_equal(xi[:6000], parts)


def test_host_sampler_partition_invariance(hostlib):
    whole = host_rvs(hostlib, "gamma", 3000, (2.5, 1.3))
    parts = np.concatenate([host_rvs(hostlib, "gamma", 1000, (2.5, 1.3), offset=o) for o in (0, 1000, 2000)])
    assert np.array_equal(whole, parts)
    a = host_rvs(hostlib, "normal", 1001, (0.0, 1.0), offset=0)
    b = host_rvs(hostlib, "normal", 500, (0.0, 1.0), offset=501)  # odd offset: second Box-Muller branch first
    assert np.array_equal(a[501:], b)


def _ziggurat_checks(x):
    """Shape of a standard-normal sample beyond what one KS statistic sees: 400 equiprobable cells (the wedges of the
    1024 layers are 0.4 % of the draws, spread over the whole range) and the tail beyond the ziggurat's r = 4.0388 (its own
    sampler: Marsaglia's exponential rejection)."""
    n = x.size
    edges = stats.norm.ppf(np.linspace(0.0, 1.0, 401))
    counts = np.histogram(x, bins=edges)[0]
    assert stats.chisquare(counts).pvalue > 1e-4
    r = 4.0388498461095045
    for lo in (r, 4.5, 5.0):
        expect = 2.0 * stats.norm.sf(lo) * n
        got = np.count_nonzero(np.abs(x) > lo)
        assert abs(got - expect) <= 5.0 * np.sqrt(expect) + 1.0, (lo, got, expect)
    assert abs(np.count_nonzero(x > 0) - 0.5 * n) <= 5.0 * np.sqrt(0.25 * n)
    tail = np.abs(x[np.abs(x) > r])
    if tail.size > 200:  # conditional law of the tail: P(|X| > t | |X| > r)
        assert stats.kstest(tail, lambda t: 1.0 - stats.norm.sf(t) / stats.norm.sf(r)).pvalue > 1e-4


def test_host_normal_ziggurat_shape(hostlib):
    x = host_rvs(hostlib, "normal", 6_000_000, (0.0, 1.0))
    _ziggurat_checks(x)
    # element 2k + 1 reads the second half of the block of element 2k; a rejected candidate moves to the element's own blocks
    a = host_rvs(hostlib, "normal", 20_001, (1.5, 0.25), offset=7)
    b = host_rvs(hostlib, "normal", 20_008, (1.5, 0.25), offset=0)
    assert np.array_equal(a, b[7:])


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_normal_ziggurat_queue_equals_per_lane_loop(gpu, hostlib):
    """k_rvs_normal_zig (scalar parameters: candidates that fail the threshold test wait in a queue in LDS) and the
    per-lane loop of k_rvs (array parameters) give bit-identical draws, for any split of the range and odd lengths;
    shape checks of the sample, tail included, at 2e8 draws."""
    from distributions_amd import normal

    n = 1_000_003
    a = normal.rvs(0.5, 2.0, size=n, random_state=13)
    b = normal.rvs(np.full(n, 0.5), np.full(n, 2.0), size=n, random_state=13)
    assert np.array_equal(a, b)
    for h in (123_456, 123_457):  # (an odd offset takes the per-lane kernel for the second part)
        c = np.concatenate([normal.rvs(0.5, 2.0, size=h, random_state=(13, 0)), normal.rvs(0.5, 2.0, size=n - h, random_state=(13, h))])
        assert np.array_equal(a, c)
    # (the host build's log differs in the last bits: the handful of tail draws may, too)
    assert np.allclose(a[:200_000], host_rvs(hostlib, "normal", 200_000, (0.5, 2.0), seed=13), rtol=1e-12, atol=0.0)
    for size in (1, 2, 63, 127, 129):
        assert np.array_equal(normal.rvs(0.5, 2.0, size=size, random_state=13), a[:size])
    # parameters as arrays of any layout take the same kernel (k_rvs_normal_zig<true>): mu + sigma z of the standard draws
    # (sigma a power of two: the fma rounds once, like numpy's product and sum)
    z = normal.rvs(0.0, 1.0, size=977 * 1024, random_state=13)
    mus = np.linspace(-3.0, 3.0, 1024)
    cols = normal.rvs(mus[None, :], 2.0, size=(977, 1024), random_state=13)
    assert np.array_equal(cols.reshape(-1), 2.0 * z + np.tile(mus, 977))
    sig = np.exp2(np.arange(977) % 5 - 2.0)
    rows = normal.rvs(0.25, sig[:, None], size=(977, 1024), random_state=13)
    assert np.array_equal(rows.reshape(-1), np.repeat(sig, 1024) * z + 0.25)
    last = normal.rvs(np.tile(mus, 977)[:-1], 2.0, size=977 * 1024 - 1, random_state=13)  # odd length: the last element
    assert np.array_equal(last, (2.0 * z + np.tile(mus, 977))[:-1])                       # takes ITS mu (per-lane launch)
    odd = normal.rvs(np.tile(mus, 977)[3:], 2.0, size=977 * 1024 - 3, random_state=(13, 3))  # per-lane kernel, arrays
    assert np.array_equal(odd, (2.0 * z + np.tile(mus, 977))[3:])
    x = normal.rvs(0.0, 1.0, size=200_000_000, random_state=14)
    assert np.isfinite(x).all() and np.abs(x).max() < 7.5
    _ziggurat_checks(x)


@pytest.mark.gpu
def test_gpu_normal_ziggurat_partial_last_trip(gpu):
    """Sizes beyond the grid cap whose last trip covers ONE pair: the wavefront that takes it has entries waiting from its
    full trips, and an open candidate of the partial trip must still reach the queue's test (the lanes past the end stay in
    the wave-wide bookkeeping).  Against the per-lane kernel (odd offset), many seeds: the case needs an open candidate
    in that one pair."""
    import torch

    from distributions_amd import _dispatch

    n = 2 * (64 * 312_500 + 1)  # 40 000 002 draws: 131 072 wavefronts, two full trips and the partial one for wavefront 50 356
    for seed in range(400):
        a = _dispatch.rvs("normal", (0.0, 1.0), size=n + 1, random_state=(seed, 0), as_torch=True)
        b = _dispatch.rvs("normal", (0.0, 1.0), size=n - 1, random_state=(seed, 1), as_torch=True)  # per-lane kernel
        assert torch.equal(a[1:n], b), seed
        assert torch.isfinite(a[n - 2:]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name, params, d", CONT, ids=[f"{c[0]}-{c[1]}" for c in CONT])
def test_gpu_rvs_continuous(gpu, hostlib, name, params, d):
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    n = 2_000_000
    x = m.rvs(*params, size=n, random_state=20260313)
    assert x.shape == (n,) and x.dtype == np.float64
    assert stats.kstest(x, d.cdf).pvalue > 1e-3
    assert moments_ok(x, d)
    assert stats.ks_2samp(x[:400_000], R.rvs(name, params, 400_000, seed=7)).pvalue > 1e-3
    # draw-by-draw agreement with the host build of the same sampler code (libm differences can
    # flip a borderline rejection in a handful of places at most)
    h = host_rvs(hostlib, name, 200_000, params)
    agree = np.abs(x[:200_000] - h) <= 1e-9 * np.maximum(1.0, np.abs(h))
    assert agree.mean() > 0.9999, agree.mean()


@pytest.mark.gpu
@pytest.mark.parametrize("name, params, d", DISC, ids=[f"{c[0]}-{c[1]}" for c in DISC])
def test_gpu_rvs_discrete(gpu, hostlib, name, params, d):
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    n = 2_000_000
    x = m.rvs(*params, size=n, random_state=20260313)
    assert x.dtype == np.int64 and x.min() >= 0
    assert chi2_pvalue(x, d) > 1e-3
    assert moments_ok(x.astype(np.float64), d)
    h = host_rvs_scalar(hostlib, name, 200_000, params)
    assert (x[:200_000] == h).mean() > 0.9999
    # parameters as arrays: the per-lane samplers (PTRS / BTRS / sequential inversion)
    xa = m.rvs(*[np.full(400_000, p, dtype=np.float64) for p in params], size=400_000, random_state=20260313)
    assert xa.dtype == np.int64 and chi2_pvalue(xa, d) > 1e-3
    assert (xa[:200_000] == host_rvs(hostlib, name, 200_000, params)).mean() > 0.9999


@pytest.mark.gpu
def test_gpu_rvs_semantics(gpu):
    from distributions_amd import binomial, gamma, normal, poisson

    # partition invariance (what makes 1/2/4/8-GPU generation identical): (seed, offset) chunks
    whole = gamma.rvs(2.5, 1.3, size=10_000, random_state=(99, 0))
    parts = np.concatenate([gamma.rvs(2.5, 1.3, size=2_500, random_state=(99, o)) for o in (0, 2500, 5000, 7500)])
    assert np.array_equal(whole, parts)
    w = normal.rvs(0.0, 1.0, size=4001, random_state=(5, 0))
    assert np.array_equal(w[1001:], normal.rvs(0.0, 1.0, size=3000, random_state=(5, 1001)))
    assert np.array_equal(w, normal.rvs(0.0, 1.0, size=4001, random_state=(5, 0)))
    # size semantics: None -> broadcast shape of the parameters; tuple; parameters per element
    mus = np.array([[1.0], [50.0], [400.0]])
    assert poisson.rvs(mus).shape == (3, 1)
    x = poisson.rvs(mus, size=(3, 20000), random_state=3)
    assert x.shape == (3, 20000)
    assert np.allclose(x.mean(axis=1), mus[:, 0], rtol=0.02)
    assert normal.rvs(0.0, 1.0).shape == ()
    with pytest.raises(ValueError):
        normal.rvs(np.zeros(3), 1.0, size=(2,))
    # numpy Generator as random_state: advances it, so consecutive calls differ; same seed repeats
    g = np.random.default_rng(1)
    a, b = normal.rvs(0.0, 1.0, size=10, random_state=g), normal.rvs(0.0, 1.0, size=10, random_state=g)
    assert not np.array_equal(a, b)
    assert np.array_equal(a, normal.rvs(0.0, 1.0, size=10, random_state=np.random.default_rng(1)))
    n_arr = np.array([5, 50, 500])
    xb = binomial.rvs(n_arr[:, None], 0.5, size=(3, 5000), random_state=4)
    assert np.all(xb <= n_arr[:, None]) and np.allclose(xb.mean(axis=1), n_arr * 0.5, rtol=0.03)


@pytest.mark.gpu
@pytest.mark.parametrize("name, params, d", DISC + [("poisson", (0.05,), stats.poisson(0.05)), ("poisson", (5000.0,), stats.poisson(5000)),
                                                    ("binomial", (2000, 0.999), stats.binom(2000, 0.999))],
                         ids=lambda v: str(v) if isinstance(v, (str, tuple)) else "")
def test_gpu_inversion_sampler(gpu, hostlib, name, params, d):
    """k_rvs_inv (scalar parameters): the host build's draws, any split of the range over calls (odd offsets: the second
    half of a Philox block first), sizes down to one element, and the law at 2e7 draws."""
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    n = 300_007
    a = m.rvs(*params, size=n, random_state=11)
    assert a.dtype == np.int64
    h = host_rvs_inv(hostlib, name, n, params, seed=11)
    assert h is not None and (a == h).mean() > 0.9999
    for cut in (123_456, 123_457):
        c = np.concatenate([m.rvs(*params, size=cut, random_state=(11, 0)), m.rvs(*params, size=n - cut, random_state=(11, cut))])
        assert np.array_equal(a, c)
    for size in (1, 2, 3, 511, 1025):
        assert np.array_equal(m.rvs(*params, size=size, random_state=11), a[:size])
        assert np.array_equal(m.rvs(*params, size=size, random_state=(11, 77)), a[77:77 + size])
    x = m.rvs(*params, size=20_000_000, random_state=12)
    assert chi2_pvalue(x, d) > 1e-4 and moments_ok(x.astype(np.float64), d)


@pytest.mark.gpu
def test_gpu_inversion_sampler_dtypes_and_declined_parameters(gpu):
    """k_rvs_inv through the C-ABI: float64 / float32 outputs and misaligned `out` (element-wise stores) carry the int64
    draws; parameter sets the tabulated path declines keep their per-lane results."""
    import ctypes

    import torch

    from distributions_amd import _lib, binomial, poisson, zi_poisson
    from distributions_amd._lib import F32, F64, I64, Operand

    L = _lib.lib()
    dev = torch.device("cuda", 0)
    n = 100_001

    def raw(name, p, odt, tdt, offset=0, shift=0):
        ops = (Operand * len(p))()
        for j, v in enumerate(p):
            ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = None, float(v), 1, 1, F64
        buf = torch.full((n + 2,), -7, dtype=tdt, device=dev)
        out = buf[shift:shift + n]
        _lib.check(L.m355_rvs(DIST[name], odt, n, len(p), ops, 99, DIST[name], offset, out.data_ptr(),
                              torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize()
        assert buf[:shift].eq(-7).all() and buf[shift + n:].eq(-7).all()  # nothing written outside [0, n)
        return out.cpu().numpy()

    for name, p in (("poisson", (100.0,)), ("binomial", (1000.0, 0.3)), ("negativebinomial", (20.0, 0.05))):
        ref = raw(name, p, I64, torch.int64)
        assert np.array_equal(ref, raw(name, p, F64, torch.float64).astype(np.int64))
        assert np.array_equal(ref, raw(name, p, F32, torch.float32).astype(np.int64))
        assert np.array_equal(ref, raw(name, p, I64, torch.int64, shift=1))  # 8-byte aligned only
        assert np.array_equal(ref[5:], raw(name, p, I64, torch.int64, offset=5)[:n - 5])
    assert poisson.rvs(0.0, size=1000, random_state=1).max() == 0
    assert np.array_equal(binomial.rvs(50, 1.0, size=1000, random_state=1), np.full(1000, 50))
    assert binomial.rvs(50, 0.0, size=1000, random_state=1).max() == 0
    assert binomial.rvs(0, 0.3, size=1000, random_state=1).max() == 0
    assert zi_poisson.rvs(0.0, 3.0, size=1000, random_state=1).max() == 0
    assert poisson.rvs(100.0, size=0).shape == (0,)


@pytest.mark.gpu
@pytest.mark.parametrize("name, params", [(c[0], c[1]) for c in DISC_BIG] + [("poisson", (1.0e6,))])
def test_gpu_compaction_equals_per_lane_loop(gpu, name, params):
    """The wavefront-compacting kernel (scalar parameters whose window is too long for k_rvs_inv) and the per-lane
    rejection loop (array parameters) give bit-identical draws: a draw depends only on (seed, stream, element, attempt)."""
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    n = 300_007
    a = m.rvs(*params, size=n, random_state=11)
    arr = [np.full(n, p, dtype=np.float64) for p in params]
    b = m.rvs(*arr, size=n, random_state=11)
    assert a.dtype == np.int64 and np.array_equal(a, b)
    # ... and any split of the range over calls (GPUs) concatenates to the same stream
    h = 123_456
    c = np.concatenate([m.rvs(*params, size=h, random_state=(11, 0)), m.rvs(*params, size=n - h, random_state=(11, h))])
    assert np.array_equal(a, c)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["poisson", "binomial"])
def test_gpu_array_parameter_compaction(gpu, hostlib, name):
    """k_rvs_compact_arr (PTRS / BTRS with one parameter set per element, n >= 4096) gives the draws of the per-lane loop
    of k_rvs (calls below 4096 elements) and of the host build, for parameters that mix every branch of the samplers."""
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    rng = np.random.default_rng(5)
    n = 200_003
    if name == "poisson":
        mu = np.exp(rng.uniform(np.log(0.05), np.log(5e5), n))
        mu[::97] = 0.0
        params = (mu,)
    else:
        nn = np.floor(np.exp(rng.uniform(0.0, np.log(1e6), n)))
        pr = rng.uniform(0.0, 1.0, n)
        pr[::89] = 0.0
        pr[1::89] = 1.0
        nn[2::89] = 0.0
        params = (nn, pr)
    a = m.rvs(*params, size=n, random_state=21)
    assert a.dtype == np.int64
    for off in (0, 1000, 77_777, n - 1500):
        cnt = min(1500, n - off)
        small = m.rvs(*[p[off:off + cnt] for p in params], size=cnt, random_state=(21, off))
        assert np.array_equal(a[off:off + cnt], small)
    for i in range(0, 3000, 300):  # the host build, one parameter set at a time
        h = host_rvs(hostlib, name, 1, tuple(float(p[i]) for p in params), seed=21, offset=i)
        assert a[i] == h[0], (i, a[i], h[0])
    # the law, at fixed parameters passed as arrays
    fixed = (np.full(400_000, 250.0),) if name == "poisson" else (np.full(400_000, 5000.0), np.full(400_000, 0.97))
    d = stats.poisson(250.0) if name == "poisson" else stats.binom(5000, 0.97)
    x = m.rvs(*fixed, size=400_000, random_state=22)
    assert chi2_pvalue(x, d) > 1e-3 and moments_ok(x.astype(np.float64), d)
    h = host_rvs(hostlib, name, 100_000, tuple(float(p[0]) for p in fixed), seed=22)
    assert (x[:100_000] == h).mean() > 0.9999


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gamma", "inversegamma", "chisquared", "scaledinversechisquared", "studentt", "halfstudentt"])
def test_gpu_array_parameter_mt_compaction(gpu, hostlib, name):
    """k_rvs_mt_arr (one Marsaglia-Tsang stage, one parameter set per element, n >= 4096) gives the draws of the per-lane
    loop of k_rvs (calls below 4096 elements): shapes from 0.05 to 1e4 (boost below 1), a few degenerate ones."""
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    rng = np.random.default_rng(6)
    n = 150_001
    shape = np.exp(rng.uniform(np.log(0.05), np.log(1e4), n))
    shape[::101] = 0.0
    shape[1::101] = np.inf
    scale = np.exp(rng.uniform(-2.0, 2.0, n))
    params = {"gamma": (shape, scale), "inversegamma": (shape, scale), "chisquared": (2.0 * shape,),
              "scaledinversechisquared": (2.0 * shape, scale), "studentt": (2.0 * shape, rng.normal(size=n), scale),
              "halfstudentt": (2.0 * shape, scale)}[name]
    with np.errstate(all="ignore"):
        a = m.rvs(*params, size=n, random_state=23)
        for off in (0, 3000, 77_777, n - 2500):
            cnt = min(2500, n - off)
            small = m.rvs(*[p[off:off + cnt] for p in params], size=cnt, random_state=(23, off))
            assert np.array_equal(a[off:off + cnt], small, equal_nan=True)
    ok = np.isfinite(shape) & (shape > 0)
    assert np.isfinite(a[ok]).all()
    # the law at fixed parameters passed as arrays, against the scalar-parameter kernel's sample
    fixed = {"gamma": (2.5, 1.3), "inversegamma": (3.0, 2.0), "chisquared": (7.0,), "scaledinversechisquared": (4.0, 1.5),
             "studentt": (5.0, 0.0, 2.0), "halfstudentt": (5.0, 2.0)}[name]
    x = m.rvs(*[np.full(400_000, v) for v in fixed], size=400_000, random_state=24)
    y = m.rvs(*fixed, size=400_000, random_state=25)
    assert stats.ks_2samp(x, y).pvalue > 1e-3
    assert np.array_equal(x, m.rvs(*fixed, size=400_000, random_state=24))  # (the same attempts as the scalar kernel)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["negativebinomial", "zi_poisson", "zi_negativebinomial", "zi_binomial", "betabinomial"])
def test_gpu_array_parameter_count_compaction(gpu, hostlib, name):
    """The compound count families with array parameters: k_rvs_zi_arr (zi_poisson / zi_binomial: inflation uniform, then
    PTRS / BTRS with the lane refill, n >= 4096) gives the draws of the per-lane loop of k_rvs, and so do the families that
    keep that loop at every n (a split-invariance check for them): base parameters on both sides of the rejection samplers'
    limits, gamma shapes on both sides of 1, degenerate values; the law at fixed parameters against the inversion kernel."""
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    rng = np.random.default_rng(12)
    n = 150_001
    psi = rng.uniform(size=n)
    psi[::89] = 0.0
    psi[1::89] = 1.0
    shape = np.exp(rng.uniform(np.log(0.05), np.log(1e3), n))
    shape[::101] = 0.0
    shape[1::101] = np.inf
    prob = rng.uniform(0.001, 1.0, n)
    prob[::103] = 1.0
    prob[1::103] = 0.0
    mu = np.exp(rng.uniform(np.log(0.05), np.log(5e4), n))
    mu[::107] = 0.0
    trials = np.floor(np.exp(rng.uniform(0.0, np.log(1e5), n)))
    trials[::109] = 0.0
    al = np.exp(rng.uniform(np.log(0.2), np.log(200.0), n))
    be = np.exp(rng.uniform(np.log(0.2), np.log(200.0), n))
    al[::113] = 0.0
    params = {"negativebinomial": (shape, prob), "zi_poisson": (psi, mu), "zi_negativebinomial": (psi, shape, prob),
              "zi_binomial": (psi, trials, prob), "betabinomial": (trials, al, be)}[name]
    with np.errstate(all="ignore"):
        x = m.rvs(*params, size=n, random_state=31)
        for off in (0, 3000, 77_777, n - 2500):
            cnt = min(2500, n - off)
            small = m.rvs(*[p[off:off + cnt] for p in params], size=cnt, random_state=(31, off))
            assert np.array_equal(x[off:off + cnt], small, equal_nan=True)
        # one array parameter, the others scalars
        fixed = {"negativebinomial": (20.0, 0.05), "zi_poisson": (0.3, 40.0), "zi_negativebinomial": (0.7, 5.0, 0.04),
                 "zi_binomial": (0.5, 400.0, 0.3), "betabinomial": (300.0, 20.0, 30.0)}[name]
        cnt = 300_000
        mixed = [np.full(cnt, fixed[0])] + list(fixed[1:])
        y = m.rvs(*mixed, size=cnt, random_state=32)
        for off in (0, 123_456):
            small = m.rvs(mixed[0][:2000], *fixed[1:], size=2000, random_state=(32, off))
            assert np.array_equal(y[off:off + 2000], small)
    z = m.rvs(*fixed, size=cnt, random_state=33)  # scalar parameters: the inversion kernel, another stream
    assert abs(y.mean() - z.mean()) < 6.0 * np.sqrt((y.var() + z.var()) / cnt)
    assert stats.ks_2samp(y, z).pvalue > 1e-4
    assert abs(np.mean(y == 0) - np.mean(z == 0)) < 6.0 * np.sqrt(0.5 / cnt)


@pytest.mark.gpu
@pytest.mark.parametrize("name, params", [("skewnormal", (0.5, 1.5, 2.0)), ("exgaussian", (0.0, 1.0, 2.0)), ("wald", (1.0, 2.0)),
                                          ("vonmises", (0.5, 2.0)), ("vonmises", (-2.0, 50.0)), ("vonmises", (0.5, 1e-6)),
                                          ("vonmises", (0.0, 1e-9)), ("rice", (1.0, 2.0)), ("halfstudentt", (5.0, 2.0)),
                                          ("halfstudentt", (0.6, 1.0)), ("polyagamma", (1.0, 0.5)), ("truncatednormal", (0.0, 1.0, -1.0, 2.0)),
                                          ("skew_studentt", (0.0, 1.0, 3.0, 4.0)), ("geometric", (0.2,)), ("kumaraswamy", (2.0, 3.0))])
def test_gpu_scalar_parameter_variants_equal_general_kernels(gpu, name, params):
    """k_rvs<D, SCALARS> / k_rvs_pairs<D, SCALARS> (all parameters scalar, read once in front of the loop; n >= 4096 for
    k_rvs) and k_rvs_vm (von Mises with the lane refill, kappa >= 1e-8) give the draws of the general kernels: the same call with one parameter passed as a constant array, and calls
    below 4096 elements at the matching offsets."""
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    n = 30_001
    a = m.rvs(*params, size=n, random_state=41)
    b = m.rvs(np.full(n, params[0]), *params[1:], size=n, random_state=41)
    assert np.array_equal(a, b, equal_nan=True)
    for off in (0, 4000, 27_502):
        cnt = min(2499, n - off)
        assert np.array_equal(a[off:off + cnt], m.rvs(*params, size=cnt, random_state=(41, off)), equal_nan=True)


@pytest.mark.gpu
def test_gpu_array_parameter_kernels_dtypes_and_lengths(gpu):
    """The array-parameter compaction kernels through the C-ABI: float32 / int64 outputs, a misaligned `out`, lengths on both sides of the 4096-element switch from the per-lane loop -- the draws of element i do
    not depend on any of it, and nothing is written outside [0, n)."""
    import ctypes

    import torch

    from distributions_amd import _lib
    from distributions_amd._lib import F32, F64, I64, Operand

    L = _lib.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    u = torch.rand(20_000, dtype=torch.float64, device=dev, generator=g)

    def raw(name, p, n, odt, tdt, shift=0, pdt=torch.float64):
        ops = (Operand * len(p))()
        keep = []
        for j, v in enumerate(p):
            if torch.is_tensor(v):
                t = v[:n].to(pdt).contiguous()
                keep.append(t)
                ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = t.data_ptr(), 0.0, 1, n, F64 if pdt == torch.float64 else F32
            else:
                ops[j].ptr, ops[j].scalar, ops[j].div, ops[j].mod, ops[j].dtype = None, float(v), 1, 1, F64
        buf = torch.full((n + 2,), -7, dtype=tdt, device=dev)
        out = buf[shift:shift + n]
        _lib.check(L.m355_rvs(DIST[name], odt, n, len(p), ops, 77, DIST[name], 0, out.data_ptr(),
                              torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize()
        assert buf[:shift].eq(-7).all() and buf[shift + n:].eq(-7).all()
        return out.cpu().numpy()

    f32 = lambda t: t.to(torch.float32).to(torch.float64)  # parameters exactly representable in float32
    cases = (("gamma", (f32(0.3 + 30.0 * u), 1.5), False), ("studentt", (f32(1.0 + 20.0 * u), 0.0, 2.0), False),
             ("beta", (f32(0.5 + 6.0 * u), 3.0), False), ("poisson", (f32(2.0 + 400.0 * u),), True),
             ("binomial", (1000.0, f32(u)), True), ("zi_poisson", (0.4, f32(2.0 + 400.0 * u)), True),
             ("zi_binomial", (f32(u), 800.0, 0.3), True))
    for name, p, discrete in cases:
        ref = raw(name, p, 20_000, F64, torch.float64)
        for n in (4095, 4096, 4097, 12_345):
            assert np.array_equal(ref[:n], raw(name, p, n, F64, torch.float64)), (name, n)
        with pytest.raises(_lib.M355Error, match="dtype"):  # parameter arrays are float64 (include/m355_dist.h)
            raw(name, p, 20_000, F64, torch.float64, pdt=torch.float32)
        assert np.array_equal(ref, raw(name, p, 20_000, F64, torch.float64, shift=1)), name
        assert np.array_equal(ref.astype(np.float32), raw(name, p, 20_000, F32, torch.float32)), name
        if discrete:
            assert np.array_equal(ref.astype(np.int64), raw(name, p, 20_000, I64, torch.int64)), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["beta", "betascaled"])
def test_gpu_array_parameter_beta_compaction(gpu, hostlib, name):
    """k_rvs_bb_arr (Cheng's BB with the constants per lane, n >= 4096) gives the draws of the per-lane loop of k_rvs:
    shapes on both sides of 1 (the gamma-ratio branch is taken on the spot), a few degenerate ones."""
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    rng = np.random.default_rng(8)
    n = 150_001
    a = np.exp(rng.uniform(np.log(0.2), np.log(500.0), n))
    b = np.exp(rng.uniform(np.log(0.2), np.log(500.0), n))
    a[::97] = 1.0
    b[1::97] = np.inf
    a[2::97] = 0.0
    lo, hi = rng.normal(size=n), 3.0 + rng.uniform(size=n)
    params = (a, b) if name == "beta" else (a, b, lo, hi)
    with np.errstate(all="ignore"):
        x = m.rvs(*params, size=n, random_state=29)
        for off in (0, 3000, 77_777, n - 2500):
            cnt = min(2500, n - off)
            small = m.rvs(*[p[off:off + cnt] for p in params], size=cnt, random_state=(29, off))
            assert np.array_equal(x[off:off + cnt], small, equal_nan=True)
    ok = np.isfinite(b) & (a > 0)
    assert np.isfinite(x[ok]).all()
    fixed = (2.0, 5.0) if name == "beta" else (2.0, 5.0, -1.0, 3.0)
    y = m.rvs(*[np.full(400_000, v) for v in fixed], size=400_000, random_state=30)
    assert np.array_equal(y, m.rvs(*fixed, size=400_000, random_state=30))  # (the same attempts as the scalar kernel)
    z = (y + 1.0) / 4.0 if name == "betascaled" else y
    assert stats.kstest(z, stats.beta(2.0, 5.0).cdf).pvalue > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name, params", [("gamma", (2.5, 1.3)), ("gamma", (1.0, 2.0)), ("beta", (2.0, 5.0)),
                                          ("studentt", (5.0, 0.0, 2.0)), ("chisquared", (7.0,)), ("inversegamma", (3.0, 2.0)),
                                          ("scaledinversechisquared", (4.0, 1.5)), ("betascaled", (2.0, 3.0, -1.0, 4.0)),
                                          ("gamma", (0.5, 1.0)), ("beta", (0.5, 0.7)), ("beta", (0.3, 4.0)), ("studentt", (1.0, 0.0, 1.0)),
                                          ("beta", (1.0, 3.0)), ("beta", (1.01, 1.02)), ("beta", (50.0, 3.0)), ("betascaled", (1.0, 0.5, -1.0, 4.0)),
                                          ("chisquared", (0.7,))])
def test_gpu_marsaglia_tsang_compaction_equals_per_lane_loop(gpu, name, params):
    """k_rvs_mt / k_rvs_bb (scalar parameters: one Philox block per attempt, lanes refilled as they finish) and the per-lane
    loop of k_rvs (array parameters) give bit-identical draws, for any split of the range over calls."""
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    n = 300_007
    a = m.rvs(*params, size=n, random_state=12)
    arr = [np.full(n, p, dtype=np.float64) for p in params]
    b = m.rvs(*arr, size=n, random_state=12)
    assert a.dtype == np.float64 and np.array_equal(a, b)
    h = 123_457
    c = np.concatenate([m.rvs(*params, size=h, random_state=(12, 0)), m.rvs(*params, size=n - h, random_state=(12, h))])
    assert np.array_equal(a, c)
    small = m.rvs(*params, size=1000, random_state=(12, 0))  # (below the kernel's threshold: the per-lane loop)
    assert np.array_equal(a[:1000], small)


FAMILY_RVS = [
    ("exponential", (0.27,), lambda: stats.expon(scale=1 / 0.27)),
    ("lognormal", (2.0, 0.5), lambda: stats.lognorm(s=0.5, scale=np.exp(2.0))),
    ("halfnormal", (3.5,), lambda: stats.halfnorm(scale=3.5)),
    ("cauchy", (-5.0, 0.5), lambda: stats.cauchy(loc=-5.0, scale=0.5)),
    ("halfcauchy", (3.5,), lambda: stats.halfcauchy(scale=3.5)),
    ("laplace", (10.0, 2.0), lambda: stats.laplace(loc=10.0, scale=2.0)),
    ("logistic", (2.5, 4.0), lambda: stats.logistic(loc=2.5, scale=4.0)),
    ("gumbel", (2.5, 3.5), lambda: stats.gumbel_r(loc=2.5, scale=3.5)),
    ("weibull", (2.0, 5.0), lambda: stats.weibull_min(c=2.0, scale=5.0)),
    ("pareto", (2.5, 2.0), lambda: stats.pareto(b=2.5, scale=2.0)),
    ("uniform", (-5.0, 5.0), lambda: stats.uniform(loc=-5.0, scale=10.0)),
    ("triangular", (0.0, 0.1, 1.0), lambda: stats.triang(c=0.1, loc=0.0, scale=1.0)),
    ("kumaraswamy", (1.00000001, 5.0), lambda: stats.beta(a=1.00000001, b=5.0)),
    ("moyal", (-5.0, 0.5), lambda: stats.moyal(loc=-5.0, scale=0.5)),
    ("asymmetriclaplace", (-1.0, 2.0, 2.0), lambda: stats.laplace_asymmetric(kappa=2.0, loc=-1.0, scale=2.0)),
    ("loglogistic", (2.0, 3.0), lambda: stats.fisk(c=3.0, scale=2.0)),
    ("inversegamma", (5.0, 2.0), lambda: stats.invgamma(a=5.0, scale=2.0)),
    ("chisquared", (5.0,), lambda: stats.chi2(df=5.0)),
    ("scaledinversechisquared", (10.0, 2.0), lambda: stats.invgamma(a=5.0, scale=10.0)),
    ("halfstudentt", (1.0, 3.5), lambda: stats.halfcauchy(scale=3.5)),
    ("halfstudentt", (5.0, 2.0), lambda: _folded(stats.t(df=5.0, scale=2.0))),
    ("betascaled", (2.0, 5.0, -1.0, 3.0), lambda: stats.beta(a=2.0, b=5.0, loc=-1.0, scale=4.0)),
    ("wald", (2.0, 10.0), lambda: stats.invgauss(mu=0.2, scale=10.0)),
    ("exgaussian", (1.0, 3.0, 3.0), lambda: stats.exponnorm(K=1.0, loc=1.0, scale=3.0)),
    ("truncatednormal", (2.0, 1.5, -1.0, 5.0), lambda: stats.truncnorm(a=-2.0, b=2.0, loc=2.0, scale=1.5)),
    ("skew_studentt", (8.0, 4.0, 0.0, 1.0), lambda: stats.jf_skew_t(a=8.0, b=4.0, loc=0.0, scale=1.0)),
    ("skewnormal", (0.5, 1.0, -2.0), lambda: stats.skewnorm(a=-2.0, loc=0.5, scale=1.0)),
    ("vonmises", (0.5, 2.0), lambda: stats.vonmises(kappa=2.0, loc=0.5)),
    ("rice", (3.0, 2.0), lambda: stats.rice(b=1.5, scale=2.0)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name, params, make", FAMILY_RVS, ids=[f[0] for f in FAMILY_RVS])
def test_gpu_rvs_families(gpu, name, params, make):
    """rvs of the closed-form / re-parameterised families: KS against the SciPy distribution the
    reference's own test compares with (p > 1e-3), determinism and partition invariance."""
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    n = 200_000
    x = m.rvs(*params, size=n, random_state=21)
    assert x.shape == (n,) and x.dtype == np.float64 and np.all(np.isfinite(x))
    xk = x
    if name == "vonmises":  # draws are wrapped to [-pi, pi] (numpy's sampler); SciPy's cdf lives on [loc - pi, loc + pi]
        assert np.all(np.abs(x) <= np.pi)
        xk = np.mod(x - params[0] + np.pi, 2 * np.pi) - np.pi + params[0]
    assert stats.kstest(xk, make().cdf).pvalue > 1e-3
    assert np.array_equal(x, m.rvs(*params, size=n, random_state=21))
    h = 77_777
    assert np.array_equal(x, np.concatenate([m.rvs(*params, size=h, random_state=(21, 0)),
                                             m.rvs(*params, size=n - h, random_state=(21, h))]))


PAIR_FAMILIES = [
    ("exponential", (0.27,)), ("cauchy", (0.5, 2.0)), ("halfcauchy", (1.5,)), ("laplace", (0.5, 2.0)), ("logistic", (0.5, 2.0)),
    ("gumbel", (0.5, 2.0)), ("weibull", (1.7, 2.0)), ("pareto", (3.0, 1.5)), ("uniform", (-1.0, 3.0)), ("triangular", (0.0, 1.0, 4.0)),
    ("kumaraswamy", (2.0, 3.0)), ("asymmetriclaplace", (0.5, 1.5, 0.7)), ("loglogistic", (2.0, 4.0)), ("bernoulli", (0.3,)),
    ("geometric", (0.1,)), ("discreteuniform", (-2.0, 5.0)), ("discreteweibull", (0.8, 2.0)), ("truncatednormal", (2.0, 1.5, -1.0, 5.0)),
    ("lognormal", (2.0, 0.5)), ("halfnormal", (3.5,)), ("moyal", (0.5, 2.0)), ("logitnormal", (-1.0, 1.0)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name, params", PAIR_FAMILIES, ids=[f[0] for f in PAIR_FAMILIES])
def test_gpu_pairs_kernel_equals_per_lane_kernel(gpu, name, params):
    """k_rvs_pairs (one Philox block -> the draws of elements 2k and 2k + 1, one 16-byte store; even first element) and
    k_rvs (odd first element: one element per thread) give the same stream, with scalar parameters and with every
    parameter an array; an odd length leaves its last element, with ITS parameters, to k_rvs."""
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    n = 100_001
    a = m.rvs(*params, size=n, random_state=(31, 0))
    b = m.rvs(*params, size=n - 1, random_state=(31, 1))
    assert a.shape == (n,) and np.array_equal(a[1:], b, equal_nan=True)
    rng = np.random.default_rng(3)
    arrs = [np.full(n, p) * (1.0 if name in ("bernoulli", "geometric", "discreteuniform", "discreteweibull", "kumaraswamy", "triangular",
                                               "truncatednormal", "pareto") else rng.uniform(0.9, 1.1, n)) for p in params]
    aa = m.rvs(*arrs, size=n, random_state=(31, 0))
    bb = m.rvs(*[v[1:] for v in arrs], size=n - 1, random_state=(31, 1))
    assert np.array_equal(aa[1:], bb, equal_nan=True)
    if all(np.all(v == v[0]) for v in arrs):
        assert np.array_equal(aa, a, equal_nan=True)
    assert np.isfinite(a.astype(np.float64)).all()


@pytest.mark.gpu
def test_gpu_rvs_logitnormal(gpu):
    from distributions_amd import logitnormal

    x = logitnormal.rvs(-1.0, 1.0, size=200_000, random_state=5)
    z = np.log(x) - np.log1p(-x)
    assert stats.kstest(z, stats.norm(-1.0, 1.0).cdf).pvalue > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name, params, make", DISC_FAMILIES, ids=_DISC_IDS)
def test_gpu_rvs_discrete_families(gpu, hostlib, name, params, make):
    """rvs of the discrete families: int64 draws, chi-square against the reference's distribution,
    draw-by-draw agreement with the host build, determinism and partition invariance."""
    import importlib

    m = importlib.import_module(f"distributions_amd.{name}")
    d = make()
    n = 1_000_000
    x = m.rvs(*params, size=n, random_state=20260313)
    assert x.shape == (n,) and x.dtype == np.int64
    assert chi2_pvalue(x, d) > 1e-3
    assert moments_ok(x.astype(np.float64), d)
    h = host_rvs_scalar(hostlib, name, 200_000, params)
    assert (x[:200_000] == h).mean() > 0.9999
    if name in INV_DISTS:  # parameters as arrays: the per-lane samplers
        xa = m.rvs(*[np.full(400_000, p, dtype=np.float64) for p in params], size=400_000, random_state=20260313)
        assert xa.dtype == np.int64 and chi2_pvalue(xa, d) > 1e-3
        assert (xa[:200_000] == host_rvs(hostlib, name, 200_000, params)).mean() > 0.9999
        for size in (1, 2, 3, 1025):
            assert np.array_equal(m.rvs(*params, size=size, random_state=(20260313, 77)), x[77:77 + size])
    k = 77_777
    assert np.array_equal(x[:200_000], np.concatenate([m.rvs(*params, size=k, random_state=(20260313, 0)),
                                                       m.rvs(*params, size=200_000 - k, random_state=(20260313, k))]))
