"""bench.py's multi-rank path on CPU: `python bench.py --gpus 2` must start its own two ranks (child
torch.distributed.run), shard config 5 over them, all-reduce the four totals and print ONE rank-0 JSON line with
n_gpus = 2.  Runs with the stand-in engine of
tests/bench_dry_run_engine.py injected through M355_BENCH_ENGINE (gloo, CPU tensors, per-rank sums from the oracle): the launcher, sharding,
collective, timing protocol and reporting code are the ones the GPU run uses; only the engine object differs."""

import json
import math
import os
import subprocess
import sys

import numpy as np

from oracle import restatement as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.pop("LOCAL_RANK", None)
    e["M355_BENCH_ENGINE"] = "tests.bench_dry_run_engine:OracleDryRunEngine"
    if env:
        e.update(env)
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, env=e, timeout=300)


def _expected_totals(world, seg):
    """What OracleDryRunEngine.draw gives every rank (seed = family seed + 1000 * rank), summed over ranks."""
    tot = {}
    for name, p, seed in (("normal", (0.3, 1.7), 1), ("gamma", (2.5, 1.3), 2), ("beta", (2.0, 5.0), 3),
                          ("studentt", (5.0, 0.0, 2.0), 4)):
        parts = []
        for rank in range(world):
            g = np.random.default_rng(seed + 1000 * rank)
            x = {"normal": lambda: g.normal(0, 1, seg), "gamma": lambda: g.gamma(2.5, 1 / 1.3, seg),
                 "beta": lambda: g.beta(2, 5, seg), "studentt": lambda: 2 * g.standard_t(5, seg)}[name]()
            parts.append(math.fsum(R.TABLE[f"{name}_logpdf"](x, *p)))
        tot[name] = math.fsum(parts)
    return tot


def test_bench_self_launches_two_ranks_and_reduces():
    r = _run("--gpus", "2", "--elems", "20000", "--shard-elems", "80000", "--sub-elems", "8000",
             "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # one line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["n_per_dist_per_gpu"] == 20000 and d["checks_ok_all_ranks"] is True
    assert math.isclose(d["value"], 3 * 20000 * 2 / (d["ms_per_step"] * 1e-3), rel_tol=1e-9)
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    sh = d["sharded_sumlogp"]
    assert sh["weak"]["elements_total"] == 2 * 80000 and sh["weak"]["elements_per_gpu"] == 80000
    assert sh["strong"]["elements_total"] == 80000 and sh["strong"]["elements_per_gpu"] == 40000
    exp = _expected_totals(2, 20000)  # weak: four segments of 80000 / 4 per rank
    for name, v in exp.items():
        assert math.isclose(sh["weak"]["totals"][name], v, rel_tol=1e-13), name
    exp = _expected_totals(2, 10000)  # strong: the first 40000 / 4 of each segment
    for name, v in exp.items():
        assert math.isclose(sh["strong"]["totals"][name], v, rel_tol=1e-13), name
    assert sh["sub_problem_check"]["ok"] and sh["sub_problem_check"]["elements"] == 8000


def test_bench_single_rank_line_has_baselines():
    r = _run("--elems", "20000", "--shard-elems", "40000", "--sub-elems", "4000", "--steps", "2",
             "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["unit"] == "evals/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port"
    assert d["cpu_baseline_all_cores"]["cores"] >= 1 and d["cpu_baseline_all_cores"]["host_cpu_count"] == os.cpu_count()
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def test_bench_rejects_world_size_mismatch():
    r = _run("--gpus", "2", env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
