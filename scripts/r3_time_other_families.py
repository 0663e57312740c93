"""Round 3: rvs of the families outside config 3 and the one-uniform list, scalar parameters, 1e9 draws (2e8 for the slow ones)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributions_amd import _dispatch

def t(f, reps=2):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cases = (("skewnormal", (0.0, 1.0, 2.0)), ("exgaussian", (0.0, 1.0, 2.0)), ("wald", (1.0, 2.0)), ("vonmises", (0.5, 2.0)),
         ("rice", (1.0, 2.0)), ("halfstudentt", (5.0, 2.0)), ("truncatednormal", (0.0, 1.0, -1.0, 2.0)),
         ("skew_studentt", (0.0, 1.0, 3.0, 4.0)), ("polyagamma", (1.0, 0.5)), ("triangular", (0.0, 0.3, 1.0)),
         ("kumaraswamy", (2.0, 3.0)), ("asymmetriclaplace", (0.0, 1.0, 2.0)), ("loglogistic", (1.0, 3.0)),
         ("logitnormal", (0.0, 1.0)), ("inversegamma", (3.0, 2.0)), ("chisquared", (7.0,)),
         ("scaledinversechisquared", (4.0, 1.5)), ("betascaled", (2.0, 5.0, -1.0, 3.0)), ("discreteuniform", (0.0, 9.0)),
         ("bernoulli", (0.3,)), ("discreteweibull", (0.8, 1.5)))
for name, p in cases:
    n = 200_000_000
    try:
        ms = t(lambda: _dispatch.rvs(name, p, size=n, random_state=3, as_torch=True))
        print(json.dumps({"fn": f"{name}.rvs{p}", "n": n, "ms_per_1e9": ms * 1e9 / n, "draws_per_s": n / ms * 1e3}), flush=True)
    except Exception as e:  # a family without rvs in the reference
        print(json.dumps({"fn": f"{name}.rvs{p}", "error": str(e)[:100]}), flush=True)
