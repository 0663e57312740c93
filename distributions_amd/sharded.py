"""Element-sharded total log-likelihood across the GPUs of one node.

One process per GPU.  Every elementwise function and rvs is independent per element, so a batch is
cut into contiguous blocks of the flat index; the only exchange step of the whole path is one
all-reduce (sum) of a few float64 totals (RCCL over xGMI when the tensors are on GPUs; 8-128 bytes,
latency bound).  Nothing else is communicated.
"""

from __future__ import annotations

from typing import Optional, Sequence, Tuple


def shard_bounds(n: int, world: int, rank: int, align: int = 256, period: Optional[int] = None) -> Tuple[int, int]:
    """[lo, hi) of rank's contiguous block of a flat array of n elements.  Interior boundaries are
    multiples of `align` (and of `period` for period-broadcast parameters, so that every shard
    starts at phase 0 of the period); the last rank takes the remainder."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q = align
    if period:
        from math import gcd

        q = align * period // gcd(align, period)
    units = n // q
    base, extra = divmod(units, world)
    lo_u = rank * base + min(rank, extra)
    hi_u = lo_u + base + (1 if rank < extra else 0)
    lo, hi = lo_u * q, hi_u * q
    if rank == world - 1:
        hi = n
    return lo, hi


def allreduce_sum_(totals, group=None):
    """In-place sum over ranks of a float64 tensor of per-rank totals (RCCL for CUDA tensors,
    gloo for CPU tensors); no-op without an initialised process group."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM, group=group)
    return totals


def total_loglik(terms: Sequence[tuple], group=None):
    """terms: sequence of (dist_name, x_local, params) with x_local this rank's shard (contiguous
    torch CUDA tensor) and scalar params.  Returns (per-term totals over all ranks, grand total) as
    float64 tensors on the device.  One fused sum-logp launch per term, then one all-reduce."""
    import torch

    from . import raw

    device = terms[0][1].device
    sums = torch.zeros(len(terms), dtype=torch.float64, device=device)
    ws = raw.sumlogp_workspace(device)
    for i, (name, x, params) in enumerate(terms):
        raw.sumlogp_into(name, "logpdf", x, params, sums[i:], ws)
    allreduce_sum_(sums, group)
    return sums, sums.sum()
