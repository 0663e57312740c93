#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): logpdf evaluations/s and % of HBM roofline for Normal/Gamma/Beta at
N = 1e8 fp64 per GPU, 1-8 GPUs, plus the sharded total log-likelihood of SURVEY.md 8(d) config 5.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process starts the N ranks itself (a child
`python -m torch.distributed.run`, started before anything here touches the GPU) and exits with the child's status;
under an external launcher (WORLD_SIZE set) it is one rank.  WORLD_SIZE != --gpus is an error.

Timed regions (each: W warm-up steps, barrier + synchronize, exactly K steps, barrier + synchronize, MAX over ranks):
  A  `value`: one step = three stored elementwise logpdf launches (Normal, Gamma, Beta; N elements each,
     16 B/element) on every rank's own shard.  No collective: the elementwise path has no exchange step.
  B  `sharded_sumlogp.weak`: config 5 -- every rank holds 1e9 elements in four equal segments
     (Normal / Gamma / Beta / StudentT, config-2 parameters, seeds offset by rank); one step = four fused
     sum-logp launches (8 B/element) + ONE all-reduce (RCCL, sum) of the four float64 totals.
  C  `sharded_sumlogp.strong`: the same with a fixed total of 1e9 elements split over the ranks.
Inputs are SURVEY.md 8(d)'s distributions drawn on the device by the library's Philox sampler (seeds 1-4 + rank).
After the timed regions: every 97th element of the stored results and a 1e7-element sub-problem of the sharded total
are checked against the CPU oracle (the checker only; nothing timed touches it), and at N = 1 the oracle's C leg is
timed on the host cores (`cpu_baseline`: 1 thread; `cpu_baseline_all_cores`: a thread pool, core count stated).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling of this access pattern ~6400
CASES = (("normal", (0.3, 1.7)), ("gamma", (2.5, 1.3)), ("beta", (2.0, 5.0)))
# (family, parameters, sampler family, sampler parameters, seed)
FAMILIES = (
    ("normal", (0.3, 1.7), "normal", (0.0, 1.0), 1),
    ("gamma", (2.5, 1.3), "gamma", (2.5, 1.3), 2),        # x ~ Gamma(shape 2.5, scale 1/1.3)
    ("beta", (2.0, 5.0), "beta", (2.0, 5.0), 3),
    ("studentt", (5.0, 0.0, 2.0), "studentt", (5.0, 0.0, 2.0), 4),  # x = 2 t_5
)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    # (torch.distributed.run's own parser sees the script's options too and rejects prefixes of its own, e.g. --n)
    ap.add_argument("--elems", "--n", dest="n", type=int, default=100_000_000, help="elements per distribution per GPU (region A)")
    ap.add_argument("--shard-elems", dest="n_shard", type=int, default=1_000_000_000,
                    help="config 5: elements per GPU (weak) and in total (strong)")
    ap.add_argument("--sub-elems", dest="n_sub", type=int, default=10_000_000,
                    help="config 5: size of the sub-problem checked against math.fsum")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sharded", action="store_true", help="skip regions B and C")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------- launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def canonical_argv(args):
    out = ["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--elems", str(args.n),
           "--shard-elems", str(args.n_shard), "--sub-elems", str(args.n_sub)]
    for flag, on in (("--no-cpu-baseline", args.no_cpu_baseline), ("--no-sharded", args.no_sharded)):
        if on:
            out.append(flag)
    return out


def self_launch(args):
    """Start `--gpus` ranks of this script under torch.distributed.run as a CHILD process (this parent has not
    initialised the GPU and never will) and return its exit status."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *canonical_argv(args)]
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------------------ engines
class GpuEngine:
    """The product path: HIP kernels through the C-ABI (distributions_amd.raw), torch for buffers / streams / RCCL."""

    name = "hip"

    def __init__(self, local_rank):
        import torch

        from distributions_amd import raw

        self.torch, self.raw = torch, raw
        assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        self.backend = "nccl"
        self.ws = None

    def draw(self, fam, n, rank):
        _name, _p, sampler, sp, seed = fam
        x = self.torch.empty(n, dtype=self.torch.float64, device=self.device)
        self.raw.rvs_into(sampler, sp, x, seed=seed + 1000 * rank)
        return x

    def empty(self, n):
        return self.torch.empty(n, dtype=self.torch.float64, device=self.device)

    def zeros(self, n):
        return self.torch.zeros(n, dtype=self.torch.float64, device=self.device)

    def logpdf_into(self, name, x, p, out):
        self.raw.elemwise_into(name, "logpdf", x, p, out)

    def sumlogp_into(self, name, x, p, slot):
        if self.ws is None:
            self.ws = self.raw.sumlogp_workspace(self.device)
        self.raw.sumlogp_into(name, "logpdf", x, p, slot, self.ws)

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def synchronize(self):
        self.torch.cuda.synchronize()


def make_engine(local_rank):
    """GpuEngine, unless M355_BENCH_ENGINE = "module:Class" names a stand-in (tests/test_bench_launcher.py injects
    tests/bench_dry_run_engine.py this way to run launcher, sharding and reporting code under gloo on a CPU box).
    A stand-in's numbers are not measurements and the JSON line says so (`dry_run`)."""
    spec = os.environ.get("M355_BENCH_ENGINE")
    if not spec:
        return GpuEngine(local_rank)
    import importlib

    mod, _, cls = spec.partition(":")
    return getattr(importlib.import_module(mod), cls)(local_rank)


# ------------------------------------------------------------------------------------------------------ timed regions
def timed_region(eng, dist, world, step, nk, steps, warmup):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks.
    `step(rec)` records rec[j] before launch j and rec[nk] after the last.  Returns (seconds, avg ms per launch)."""
    def sync():
        if world > 1:
            dist.barrier()
        eng.synchronize()

    for _ in range(warmup):
        step(None)
    events = [[eng.event() for _ in range(nk + 1)] for _ in range(steps)]
    sync()
    t0 = time.perf_counter()
    for s in range(steps):
        step(events[s])
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = eng.torch.tensor([elapsed], dtype=eng.torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kdur = [sum(ev[j].elapsed_time(ev[j + 1]) for ev in events) / steps for j in range(nk)]
    return elapsed, kdur


def region_stored(eng, dist, world, rank, args):
    """Region A: three stored logpdf launches per step on this rank's own N-element shards."""
    fams = [f for f in FAMILIES if f[0] in dict(CASES)]
    xs = {f[0]: eng.draw(f, args.n, rank) for f in fams}
    out = eng.empty(args.n)

    def step(rec):
        for j, (name, p) in enumerate(CASES):
            if rec is not None:
                rec[j].record()
            eng.logpdf_into(name, xs[name], p, out)
        if rec is not None:
            rec[len(CASES)].record()

    elapsed, kdur = timed_region(eng, dist, world, step, len(CASES), args.steps, args.warmup)
    return xs, out, elapsed, kdur


def sharded_region(eng, dist, world, rank, n_local, args, xs=None):
    """Regions B / C: this rank's n_local elements in four equal segments; four fused sum-logp launches and one
    all-reduce of the four totals per step."""
    seg = n_local // len(FAMILIES)
    if xs is None:
        xs = [eng.draw(f, seg, rank) for f in FAMILIES]
    else:
        xs = [x[:seg] for x in xs]
    sums = eng.zeros(len(FAMILIES))
    nk = len(FAMILIES) + 1

    def step(rec):
        for j, f in enumerate(FAMILIES):
            if rec is not None:
                rec[j].record()
            eng.sumlogp_into(f[0], xs[j], f[1], sums[j:])
        if rec is not None:
            rec[len(FAMILIES)].record()
        if world > 1:
            dist.all_reduce(sums)  # the path's only exchange step: 32 bytes
        if rec is not None:
            rec[nk].record()

    elapsed, kdur = timed_region(eng, dist, world, step, nk, args.steps, args.warmup)
    eng.synchronize()
    totals = [float(v) for v in sums.cpu()]
    per_step = elapsed / args.steps
    elems = seg * len(FAMILIES) * world
    res = {
        "elements_total": elems, "elements_per_gpu": seg * len(FAMILIES), "ms_per_step": 1e3 * per_step,
        "evals_per_s": elems / per_step, "GB_per_s_algorithmic_per_gpu": 8.0 * seg * len(FAMILIES) / per_step / 1e9,
        "frac_of_hbm_peak_per_gpu": 8.0 * seg * len(FAMILIES) / per_step / 1e9 / HBM_PEAK_GBS,
        "kernels_avg_ms": {f"{f[0]}.sum_logpdf": kdur[j] for j, f in enumerate(FAMILIES)},
        "allreduce_avg_ms": kdur[len(FAMILIES)] if world > 1 else 0.0,
        "totals": {f[0]: totals[j] for j, f in enumerate(FAMILIES)}, "total_loglik": math.fsum(totals),
    }
    return xs, res


# ------------------------------------------------------------------------------------------- checks against the oracle
def check_stored(eng, xs, out, args):
    """Every 97th element of each stored logpdf result against the oracle: max of |a-b| / max(1, |b|)."""
    import numpy as np

    from oracle import restatement as R

    worst = {}
    for name, p in CASES:
        eng.logpdf_into(name, xs[name], p, out)
        eng.synchronize()
        got = out[::97].cpu().numpy()
        exp = R.TABLE[f"{name}_logpdf"](xs[name][::97].cpu().numpy(), *p)
        worst[name] = float(np.max(np.abs(got - exp) / np.maximum(1.0, np.abs(exp))))
    return {"points_per_dist": int(got.size), "stride": 97, "max_err": worst, "tolerance": 1e-12,
            "ok": all(v <= 1e-12 for v in worst.values())}


def check_sharded_total(eng, dist, world, rank, xs, args):
    """The sharded total of a sub-problem (n_sub elements over all ranks, four families) against math.fsum of the
    oracle's terms; tolerance 1e-12 * sum |term_i|."""
    import numpy as np

    from oracle import restatement as R

    seg = max(1, min(args.n_sub // (world * len(FAMILIES)), min(x.numel() for x in xs)))
    sums = eng.zeros(len(FAMILIES))
    for j, f in enumerate(FAMILIES):
        eng.sumlogp_into(f[0], xs[j][:seg], f[1], sums[j:])
    host = eng.torch.zeros(2, dtype=eng.torch.float64)
    terms_sum, terms_abs = [], []
    for j, f in enumerate(FAMILIES):
        t = R.TABLE[f"{f[0]}_logpdf"](xs[j][:seg].cpu().numpy(), *f[1])
        terms_sum.append(math.fsum(t))
        terms_abs.append(float(np.abs(t).sum()))
    host[0], host[1] = math.fsum(terms_sum), math.fsum(terms_abs)
    ref = host.to(eng.device)
    if world > 1:
        dist.all_reduce(sums)
        dist.all_reduce(ref)
    eng.synchronize()
    got = math.fsum(float(v) for v in sums.cpu())
    exact, scale = (float(v) for v in ref.cpu())
    return {"elements": seg * len(FAMILIES) * world, "total": got, "fsum": exact, "abs_err": abs(got - exact),
            "tolerance": 1e-12 * scale, "ok": abs(got - exact) <= 1e-12 * scale}


def pmc_traffic(dist_index):
    """HBM bytes per launch of the elementwise kernel of CASES[dist_index], REPLAYED from the newest committed
    profiles/rNN_hbm_traffic.json (separate rocprofv3 --pmc passes of this command; counters cannot be read from
    inside this process).  None if no profile has been committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        data = json.load(f)
    key = f"k_scalar<{dist_index}, 0, false, true>"
    for name, v in data.items():
        if key in name:
            return v["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    return None, None


def cpu_baselines(n_cpu, repeats):
    """The oracle's C leg (oracle/elemwise_c.c) on the host: the three stored logpdf passes of one region-A step on
    n_cpu elements per distribution; (i) one thread -- what PyTensor's fused Elemwise loop uses by default,
    (ii) all the cores this job may use, chunks over a thread pool (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    from oracle import c_leg

    L = c_leg.load()
    rng = np.random.default_rng(1)
    data = {"normal": rng.normal(0, 1, n_cpu), "gamma": rng.gamma(2.5, 1 / 1.3, n_cpu), "beta": rng.beta(2, 5, n_cpu)}
    out = np.empty(n_cpu)

    def one_thread():
        for name, p in CASES:
            c_leg.logpdf(L, name, data[name], *p, out=out)

    host_cpus = os.cpu_count() or 1
    cores = host_cpus
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):  # the box exports its per-job CPU share there
        if os.environ.get(var, "").isdigit():
            cores = min(cores, int(os.environ[var]))
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    bounds = np.linspace(0, n_cpu, cores + 1).astype(np.int64)

    def all_cores(pool):
        for name, p in CASES:
            x = data[name]
            list(pool.map(lambda k: c_leg.logpdf(L, name, x[bounds[k]:bounds[k + 1]], *p, out=out[bounds[k]:bounds[k + 1]]),
                          range(cores)))

    def best(fn):
        b = float("inf")
        for _ in range(repeats):
            t0 = time.perf_counter()
            fn()
            b = min(b, time.perf_counter() - t0)
        return 3 * n_cpu / b

    v1 = best(one_thread)
    with ThreadPoolExecutor(cores) as pool:
        vall = best(lambda: all_cores(pool))
    sample = f"oracle/elemwise_c.c: the 3 stored logpdf passes on {n_cpu} elements per distribution, best of {repeats}"
    return ({"value": v1, "unit": "evals/s", "cores": 1, "kind": "port", "sample": sample},
            {"value": vall, "unit": "evals/s", "cores": cores, "kind": "port", "sample": sample + f", {cores} threads",
             "host_cpu_count": host_cpus})


# --------------------------------------------------------------------------------------------------------------- rank
def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` "
                         "(it launches its own ranks) or under torch.distributed.run with --nproc-per-node N")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    eng = make_engine(local)
    dry = eng.name != GpuEngine.name
    import torch.distributed as dist

    if world > 1:
        if eng.backend == "nccl":
            dist.init_process_group("nccl", device_id=eng.device)
        else:
            dist.init_process_group("gloo")
    n = args.n
    xs, out, elapsed, kdur = region_stored(eng, dist, world, rank, args)
    stored_check = check_stored(eng, xs, out, args)
    del xs, out
    sharded = None
    if not args.no_sharded:
        seg_xs, weak = sharded_region(eng, dist, world, rank, args.n_shard, args)
        _, strong = sharded_region(eng, dist, world, rank, args.n_shard // world, args, xs=seg_xs)
        sub = check_sharded_total(eng, dist, world, rank, seg_xs, args)
        del seg_xs
        sharded = {"workload": "SURVEY.md 8(d) config 5: Normal/Gamma/Beta/StudentT segments, fused sum-logp, one "
                               "all-reduce of 4 float64 per step", "weak": weak, "strong": strong, "sub_problem_check": sub}
    ok = eng.torch.tensor([1.0 if (stored_check["ok"] and (sharded is None or sharded["sub_problem_check"]["ok"])) else 0.0],
                          dtype=eng.torch.float64, device=eng.device)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    all_ok = bool(ok.item() == 1.0)

    if rank == 0:
        per_step = elapsed / args.steps
        names = [f"{c[0]}.logpdf" for c in CASES]
        dom = max(range(len(CASES)), key=lambda j: kdur[j])  # slowest elementwise kernel
        achieved = 16.0 * n / (kdur[dom] * 1e-3) / 1e9
        traffic, traffic_file = pmc_traffic(dom) if (n == 100_000_000 and not dry) else (None, None)
        line = {
            "metric": "logpdf evals/sec (Normal/Gamma/Beta, N=1e8 fp64 per GPU, stored elementwise results)",
            "value": 3.0 * n * world / per_step,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[1] metric slice: Normal(0.3,1.7) / Gamma(2.5,1.3) / Beta(2,5) logpdf, "
                            "scalar params, result stored (16 B/elem); inputs x~N(0,1) / Gamma(2.5, scale 1/1.3) / "
                            "Beta(2,5), Philox seeds 1-3 (+1000 per rank)",
                "n_per_dist_per_gpu": n,
                "evals_per_step_per_gpu": 3 * n,
                "parallelism": f"element-sharded x{world}, no collective in the elementwise path" if world > 1 else "single GPU",
                "engine": eng.name,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": names[dom],
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": (f"replayed from {traffic_file}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                   "command (reads = 2 x FETCH_SIZE on gfx950); not measured in this run") if traffic else None,
                "algorithmic_bytes_per_launch": 16.0 * n,
                "avg_launch_ms": kdur[dom],
            },
            "kernels": {names[j]: {"avg_ms": kdur[j], "GB/s": 16.0 * n / (kdur[j] * 1e-3) / 1e9, "evals/s": n / (kdur[j] * 1e-3)}
                        for j in range(len(CASES))},
            "parity_spot_check": stored_check,
            "sharded_sumlogp": sharded,
            "checks_ok_all_ranks": all_ok,
        }
        if dry:
            line["dry_run"] = f"stand-in engine {os.environ['M355_BENCH_ENGINE']}: numbers are NOT measurements"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["cpu_baseline_all_cores"] = cpu_baselines(10_000_000 if not dry else 100_000, repeats=5)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not all_ok:
        raise SystemExit("bench.py: a parity check against the oracle failed (see parity_spot_check / sub_problem_check)")


def make_inputs(torch, n, device, seed=0):
    """SURVEY.md 8(d) config-2 inputs (Philox draws on the device) for bench_configs.py: dict name -> tensor."""
    from distributions_amd import raw

    xs = {}
    for name, _p, sampler, sp, s in FAMILIES:
        xs[name] = torch.empty(n, dtype=torch.float64, device=device)
        raw.rvs_into(sampler, sp, xs[name], seed=s + seed)
    return xs


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    run_rank(args)


if __name__ == "__main__":
    main()
