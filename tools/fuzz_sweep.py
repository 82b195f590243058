"""Developer tool: a longer run of tests/test_gpu_fuzz.py's generator (python tools/fuzz_sweep.py [first_seed] [count])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, oracle
from oracle import agd_oracle
from tests.helpers import NP_DT, RTOL, relerr, torch_args
from tests.test_gpu_fuzz import _case
from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
worst = {"f32": 0.0, "f64": 0.0}; bad = 0
for seed in range(first, first + count):
    p, pm, entries, col_proj, gamma, lam, dn = _case(seed)
    f = MatchingSolverDualObjectiveFunction(torch_args(p, dn, pm, "cuda:0"), gamma=gamma)
    td = torch.float32 if dn == "f32" else torch.float64
    res = f.calculate(torch.from_numpy(lam).to(td).to("cuda:0"), save_primal=True)
    ax, obj0, ssq, x = oracle.matching_calculate(p["m"], p["n"], p["colptr"], p["rowidx"], p["a"], p["c"], lam, gamma, entries, col_proj=col_proj, dtype=NP_DT[dn])
    grad = agd_oracle.epilogue(ax, obj0, ssq, lam, p["b"], gamma, NP_DT[dn])[0]
    e = max(relerr(res.primal_var.cpu().numpy(), x), relerr(res.dual_gradient.cpu().numpy(), grad))
    worst[dn] = max(worst[dn], e)
    if e > RTOL[dn] * 2:
        bad += 1
        print("MISMATCH seed", seed, dn, p["m"], p["n"], int(p["colptr"][-1]), e, f.info())
print("seeds", first, "..", first + count - 1, "mismatches", bad, "worst rel err", worst)
