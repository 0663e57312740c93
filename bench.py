#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): logpdf evaluations/s and % of HBM roofline for
Normal/Gamma/Beta at N = 1e8 fp64 per GPU, 1-8 GPUs.

One step = one pass of the hot path over one batch of synthetic, HBM-resident input on each GPU:
  3 elementwise logpdf kernels (Normal, Gamma, Beta; N elements each, result stored: 16 B/element)
  3 fused sum-logp passes of the same data (8 B/element) + their fixed-order final reduction
  1 all-reduce (RCCL, sum) of the 3 float64 totals when more than one GPU takes part
so a step evaluates 6*N log-densities per GPU.  Weak scaling: every rank owns its own N-element
shard (seed offset by rank); no data-path collective besides that 24-byte all-reduce.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the slowest
elementwise kernel (HIP events around every launch of the timed region, on the launch stream) and
`cpu_baseline` (the oracle's C leg, 1 thread, bounded sample) at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290
CASES = (("normal", (0.3, 1.7)), ("gamma", (2.5, 1.3)), ("beta", (2.0, 5.0)))


def make_inputs(torch, n, device, seed):
    """SURVEY.md 8(d) config-2 input distributions, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    xs = {}
    xs["normal"] = torch.randn(n, dtype=torch.float64, device=device, generator=g)
    u = torch.rand(n, dtype=torch.float64, device=device, generator=g)
    # Gamma(2.5, rate 1.3) via Wilson-Hilferty-free route: sum of exponentials is not needed for a
    # benchmark input; use Gamma(2,.)+Gamma(0.5,.) = -log(u1 u2) + z^2/2, exact in distribution.
    u2 = torch.rand(n, dtype=torch.float64, device=device, generator=g)
    z = torch.randn(n, dtype=torch.float64, device=device, generator=g)
    xs["gamma"] = (-(torch.log(u.clamp_min(1e-300)) + torch.log(u2.clamp_min(1e-300))) + 0.5 * z * z) / 1.3
    del u2, z
    # Beta(2,5) = G2/(G2+G5) with integer-shape gammas from uniforms
    def gint(k):
        acc = torch.zeros(n, dtype=torch.float64, device=device)
        for _ in range(k):
            acc -= torch.log(torch.rand(n, dtype=torch.float64, device=device, generator=g).clamp_min(1e-300))
        return acc
    a, b = gint(2), gint(5)
    xs["beta"] = a / (a + b)
    del a, b, u
    return xs


def pmc_traffic(dist_index):
    """HBM bytes per launch of the elementwise kernel of CASES[dist_index], from the newest
    profiles/rNN_hbm_traffic.json (written by tools/summarize_profile.py from separate
    rocprofv3 --pmc passes of this same command); None if no profile has been committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        data = json.load(f)
    key = f"k_scalar<{dist_index}, 0, false, true>"
    for name, v in data.items():
        if key in name:
            return v["hbm_bytes_per_launch"]
    return None


def cpu_baseline(n_cpu, repeats):
    """The oracle's C leg (oracle/elemwise_c.c) on one host thread: same 6 evaluations per element
    (3 stored logpdf passes + 3 logpdf-and-sum passes) as one GPU step, on n_cpu elements."""
    import numpy as np

    from oracle import c_leg

    L = c_leg.load()
    rng = np.random.default_rng(1)
    data = {
        "normal": rng.normal(0, 1, n_cpu),
        "gamma": rng.gamma(2.5, 1 / 1.3, n_cpu),
        "beta": rng.beta(2, 5, n_cpu),
    }
    out = np.empty(n_cpu)
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        for name, p in CASES:
            c_leg.logpdf(L, name, data[name], *p, out=out)          # elementwise, stored
            c_leg.logpdf(L, name, data[name], *p, out=out)          # pt.sum(logpdf(...)): loop + reduce
            L.oc_sum(n_cpu, out.ctypes.data_as(c_leg._dp))
        best = min(best, time.perf_counter() - t0)
    return 6 * n_cpu / best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=100_000_000, help="elements per distribution per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from distributions_amd import raw

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert world == max(args.gpus, 1) or world == 1, "launch with torch.distributed.run for --gpus > 1"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    n = args.n
    xs = make_inputs(torch, n, device, seed=20260313 + rank)
    out = torch.empty(n, dtype=torch.float64, device=device)
    sums = torch.zeros(3, dtype=torch.float64, device=device)
    ws = raw.sumlogp_workspace(device)
    nk = 2 * len(CASES)
    events = None

    def step(rec=None):
        j = 0
        for name, p in CASES:
            if rec is not None:
                rec[j].record()
            raw.elemwise_into(name, "logpdf", xs[name], p, out)
            j += 1
        for i, (name, p) in enumerate(CASES):
            if rec is not None:
                rec[j].record()
            raw.sumlogp_into(name, "logpdf", xs[name], p, sums[i:], ws)
            j += 1
        if rec is not None:
            rec[j].record()
        if world > 1:
            dist.all_reduce(sums)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(nk + 1)] for _ in range(args.steps)]
    sync()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(events[s])
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel average durations (ms) from the events of the timed region
    kdur = [0.0] * nk
    for ev in events:
        for j in range(nk):
            kdur[j] += ev[j].elapsed_time(ev[j + 1])
    kdur = [d / args.steps for d in kdur]
    names = [f"{c[0]}.logpdf" for c in CASES] + [f"{c[0]}.sum_logpdf" for c in CASES]
    total_ll = float(sums.sum().item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = 6.0 * n * world / (elapsed / args.steps)
        dom = max(range(len(CASES)), key=lambda j: kdur[j])  # slowest elementwise kernel
        achieved = 16.0 * n / (kdur[dom] * 1e-3) / 1e9
        traffic = pmc_traffic(dom) if n == 100_000_000 else None
        line = {
            "metric": "logpdf evals/sec (Normal/Gamma/Beta, N=1e8 fp64 per GPU: elementwise + fused sum-logp)",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[1] metric slice: Normal(0.3,1.7)/Gamma(2.5,1.3)/Beta(2,5) "
                            "logpdf, scalar params, stored (16 B/elem) + fused sum-logp (8 B/elem)",
                "n_per_dist_per_gpu": n,
                "evals_per_step_per_gpu": 6 * n,
                "parallelism": f"element-sharded x{world}, all-reduce of 3 float64" if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": names[dom],
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                  "profiles/*_hbm_traffic.json (reads = 2 x FETCH_SIZE on gfx950)" if traffic else None,
                "algorithmic_bytes_per_launch": 16.0 * n,
                "avg_launch_ms": kdur[dom],
            },
            "kernels": {
                names[j]: {"avg_ms": kdur[j], "GB/s": (16.0 if j < len(CASES) else 8.0) * n / (kdur[j] * 1e-3) / 1e9,
                           "evals/s": n / (kdur[j] * 1e-3)}
                for j in range(nk)
            },
            "total_loglik": total_ll,
        }
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = 10_000_000
            v = cpu_baseline(n_cpu, repeats=5)
            line["cpu_baseline"] = {
                "value": v, "unit": "evals/s", "cores": 1, "kind": "port",
                "sample": f"oracle/elemwise_c.c, same 6 passes on {n_cpu} elements per distribution, best of 5",
            }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
