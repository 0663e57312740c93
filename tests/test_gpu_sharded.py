"""The sharded total log-likelihood (distributions_amd/sharded.py) on the GPU: the
config-5 mix (Normal / Gamma / Beta / StudentT segments, config-2 parameters) through sharded.total_loglik against
math.fsum of the oracle's terms at 1e-12 * sum |term| -- (i) one rank without a process group, (ii) one rank inside an
initialised RCCL group of size 1 (the all-reduce really goes through RCCL), (iii) two ranks sharing the one GPU of the
box, each summing its own shard_bounds block with the kernels, totals exchanged by an all-reduce over gloo (RCCL
wants one device per rank; the collective is the same call in sharded.allreduce_sum_)."""

import math
import os
import socket

import numpy as np
import pytest

from oracle import restatement as R

pytestmark = pytest.mark.gpu

MIX = (("normal", (0.3, 1.7)), ("gamma", (2.5, 1.3)), ("beta", (2.0, 5.0)), ("studentt", (5.0, 0.0, 2.0)))


def _draw(seed, n):
    g = np.random.default_rng(seed)
    return {"normal": g.normal(0, 1, n), "gamma": g.gamma(2.5, 1 / 1.3, n), "beta": g.beta(2, 5, n),
            "studentt": 2 * g.standard_t(5, n)}


def _expected(xs):
    tot, scale = [], []
    for name, p in MIX:
        t = R.TABLE[f"{name}_logpdf"](xs[name], *p)
        tot.append(math.fsum(t))
        scale.append(float(np.abs(t).sum()))
    return tot, scale


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_total_loglik_single_rank(gpu):
    import torch

    from distributions_amd import sharded

    n = 2_500_003
    xs = _draw(41, n)
    terms = [(name, torch.from_numpy(xs[name]).to(gpu), p) for name, p in MIX]
    sums, total = sharded.total_loglik(terms)
    exp, scale = _expected(xs)
    got = sums.cpu().numpy()
    for j, (name, _p) in enumerate(MIX):
        assert abs(got[j] - exp[j]) <= 1e-12 * scale[j], (name, got[j], exp[j])
    assert abs(float(total) - math.fsum(exp)) <= 1e-12 * sum(scale)


def test_total_loglik_inside_rccl_group_of_one(gpu):
    import torch
    import torch.distributed as dist

    from distributions_amd import sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        n = 300_001
        xs = _draw(42, n)
        terms = [(name, torch.from_numpy(xs[name]).to(gpu), p) for name, p in MIX]
        sums, total = sharded.total_loglik(terms)
        t = torch.ones(4, dtype=torch.float64, device=gpu)
        dist.all_reduce(t)  # (the group is live: RCCL runs the collective even at size 1)
        exp, scale = _expected(xs)
        assert torch.equal(t, torch.ones_like(t))
        assert all(abs(g - e) <= 1e-12 * s for g, e, s in zip(sums.cpu().numpy(), exp, scale))
        assert abs(float(total) - math.fsum(exp)) <= 1e-12 * sum(scale)
    finally:
        dist.destroy_process_group()


def _rank(rank, world, port, n, out_dir):
    import torch
    import torch.distributed as dist

    from distributions_amd import sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    xs = _draw(43, n)  # the same full batch on every rank; each takes its own block
    lo, hi = sharded.shard_bounds(n, world, rank)
    terms = [(name, torch.from_numpy(xs[name][lo:hi]).to(dev), p) for name, p in MIX]
    sums, total = sharded.total_loglik(terms)
    if rank == 0:
        np.save(os.path.join(out_dir, "tot.npy"), np.append(sums.cpu().numpy(), float(total)))
    dist.destroy_process_group()


def test_total_loglik_two_ranks_on_one_gpu(gpu, tmp_path):
    import torch.multiprocessing as mp

    n = 1_000_003
    mp.spawn(_rank, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "tot.npy")
    exp, scale = _expected(_draw(43, n))
    for j, (name, _p) in enumerate(MIX):
        assert abs(got[j] - exp[j]) <= 1e-12 * scale[j], (name, got[j], exp[j])
    assert abs(got[4] - math.fsum(exp)) <= 1e-12 * sum(scale)
