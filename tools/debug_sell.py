"""developer aid: first column where the sliced and the window-tile paths disagree (tests/test_gpu_sell.py shapes)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_gpu_sell import _objective, _ragged
from dualip_amd.projections import create_projection_map
import oracle
kind = sys.argv[1] if len(sys.argv) > 1 else "simplex_eq"
p = _ragged(3)
n, m = p["n"], p["m"]
pm = create_projection_map(kind, {"z": 1.5}, n)
f, f0 = _objective(p, "f64", pm, 0.05), _objective(p, "f64", pm, 0.05, sell=False)
lam = torch.zeros(m, dtype=torch.float64, device="cuda:0")
x = f.calculate(lam, 0.05, save_primal=True).primal_var.cpu().numpy().copy()
x0 = f0.calculate(lam, 0.05, save_primal=True).primal_var.cpu().numpy().copy()
_, _, _, xo = oracle.matching_calculate(m, n, p["colptr"], p["rowidx"], p["a"], p["c"], lam.cpu().numpy(), 0.05, [(kind, {"z": 1.5})])
print("sell vs oracle", np.abs(x - xo).max(), "window vs oracle", np.abs(x0 - xo).max())
bad = np.nonzero(np.abs(x - xo) > 1e-9)[0]
cols = np.searchsorted(p["colptr"], bad, side="right") - 1
lens = np.diff(p["colptr"])
print("bad elements", len(bad), "bad columns", len(set(cols)), "their lengths", sorted(set(lens[cols]))[:40])
for j in list(dict.fromkeys(cols))[:3]:
    k0, k1 = p["colptr"][j], p["colptr"][j + 1]
    v = -(p["c"][k0:k1]) / 0.05
    print("col", j, "len", k1 - k0, "v", np.round(v, 4), "\n  sell", np.round(x[k0:k1], 4), "\n  want", np.round(xo[k0:k1], 4))
