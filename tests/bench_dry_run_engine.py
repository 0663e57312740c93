"""Stand-in engine for tests/test_bench_launcher.py (bench.py loads it when M355_BENCH_ENGINE names it): no kernels;
CPU tensors, gloo, per-rank sums from the CPU oracle, so that bench.py's launcher, sharding, collective, timing protocol
and reporting code can run on a box without GPUs.  Test infrastructure only; never a measurement."""
import math
import time


class OracleDryRunEngine:
    """TEST HOOK ONLY (--dry-run-cpu): no kernels; sums come from the CPU oracle so that launcher, sharding,
    collective and reporting code can run under gloo on a box without GPUs.  Never used for a measurement."""

    name = "oracle-dry-run"

    def __init__(self, local_rank):
        import numpy as np
        import torch

        from oracle import restatement as R

        self.torch, self.np, self.R = torch, np, R
        self.device = torch.device("cpu")
        self.backend = "gloo"

    def draw(self, fam, n, rank):
        name, _p, _s, _sp, seed = fam
        g = self.np.random.default_rng(seed + 1000 * rank)
        x = {"normal": lambda: g.normal(0, 1, n), "gamma": lambda: g.gamma(2.5, 1 / 1.3, n),
             "beta": lambda: g.beta(2, 5, n), "studentt": lambda: 2 * g.standard_t(5, n)}[name]()
        return self.torch.from_numpy(x)

    def empty(self, n):
        return self.torch.empty(n, dtype=self.torch.float64)

    def zeros(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64)

    def logpdf_into(self, name, x, p, out):
        out.copy_(self.torch.from_numpy(self.R.TABLE[f"{name}_logpdf"](x.numpy(), *p)))

    def sumlogp_into(self, name, x, p, slot):
        slot[0] = math.fsum(self.R.TABLE[f"{name}_logpdf"](x.numpy(), *p))

    class _Ev:
        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return 1e3 * (other.t - self.t)

    def event(self):
        return self._Ev()

    def synchronize(self):
        pass
