"""Multi-process (gloo, world_size 2 and 3) tests of the sharded path's host logic on CPU: the
partition covers the batch exactly once with aligned interior boundaries, and the all-reduced
per-rank totals equal the unsharded total.  Per-rank partial sums are produced by the oracle here
(the checker), because the kernels need a GPU; the GPU leg of the same path is bench.py."""

import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from distributions_amd.sharded import allreduce_sum_, shard_bounds
from oracle import restatement as R


def test_shard_bounds_cover_and_align():
    for n in (0, 1, 255, 256, 1000, 100_000_000, 100_000_003):
        for world in (1, 2, 3, 4, 8):
            for period in (None, 1024, 37):
                spans = [shard_bounds(n, world, r, 256, period) for r in range(world)]
                assert spans[0][0] == 0 and spans[-1][1] == n
                q = 256 if not period else 256 * period // math.gcd(256, period)
                for (lo, hi), (lo2, _hi2) in zip(spans, spans[1:]):
                    assert hi == lo2 and lo <= hi and hi % q == 0
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = np.random.default_rng(123).gamma(2.5, 1 / 1.3, n)  # same full batch on every rank
    z = np.random.default_rng(124).normal(0, 1, n)
    lo, hi = shard_bounds(n, world, rank)
    local = torch.tensor(
        [math.fsum(R.gamma_logpdf(x[lo:hi], 2.5, 1.3)), math.fsum(R.normal_logpdf(z[lo:hi], 0.3, 1.7))],
        dtype=torch.float64,
    )
    allreduce_sum_(local)
    if rank == 0:
        np.save(os.path.join(out_dir, "tot.npy"), local.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_allreduce_totals_gloo(tmp_path, world):
    n = 200_003
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "tot.npy")
    x = np.random.default_rng(123).gamma(2.5, 1 / 1.3, n)
    z = np.random.default_rng(124).normal(0, 1, n)
    exp = [math.fsum(R.gamma_logpdf(x, 2.5, 1.3)), math.fsum(R.normal_logpdf(z, 0.3, 1.7))]
    assert np.allclose(got, exp, rtol=1e-13, atol=0)


def test_allreduce_without_group_is_noop():
    t = torch.tensor([1.5, 2.5], dtype=torch.float64)
    assert torch.equal(allreduce_sum_(t.clone()), t)
