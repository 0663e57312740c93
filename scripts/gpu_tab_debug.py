import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from distributions_amd import _lib, studentt
L = _lib.lib()
dev = torch.device("cuda:0")
n = 8_000_000
q = torch.rand(n, dtype=torch.float64, device=dev)
for params in ((5.0, 0.0, 2.0), (30.0, 0.0, 1.0), (300.0, -2.0, 1.0)):
    L.m355_set_tab_min_n(2**62); exact = studentt.ppf(q, *params)
    L.m355_set_tab_min_n(-1); tab = studentt.ppf(q, *params)
    same = (tab == exact).cpu().numpy()
    w = np.log(q.cpu().numpy() / (1 - q.cpu().numpy()))
    cell = np.floor((w + 20.723265835946928) / (2 * 20.723265835946928 / 64)).astype(int)
    frac = np.array([same[cell == c].mean() if (cell == c).any() else np.nan for c in range(64)])
    print(params, "miss-like cells (identical frac > 0.5):", [c for c in range(64) if frac[c] > 0.5], " overall identical", same.mean())
    x = (params[1] + params[2] * torch.distributions.StudentT(params[0]).sample((n,))).double().to(dev)
    L.m355_set_tab_min_n(2**62); exact = studentt.logcdf(x, *params)
    L.m355_set_tab_min_n(-1); tab = studentt.logcdf(x, *params)
    same = (tab == exact).cpu().numpy()
    z = ((x - params[1]) / params[2]).cpu().numpy()
    ww = np.arcsinh(z)
    edges = np.quantile(ww, np.linspace(0, 1, 41))
    fr = [same[(ww >= edges[i]) & (ww < edges[i + 1])].mean() for i in range(40)]
    print("  logcdf identical frac by w-quantile bin:", np.round(fr, 2))
