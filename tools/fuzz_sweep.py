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
    if e > RTOL[dn] * 2 and dn == "f32":
        # ill-conditioned in float32 (u ~ 1e4 against z = 1)?  compare with the oracle's own f32-vs-f64 spread on the same f32 inputs
        f32 = lambda v: np.asarray(v, dtype=np.float32)
        x64 = oracle.matching_calculate(p["m"], p["n"], p["colptr"], p["rowidx"], f32(p["a"]), f32(p["c"]), f32(lam), gamma, entries, col_proj=col_proj, dtype=np.float64)[3]
        spread = relerr(x, x64)
        if e <= 4 * spread:
            print("note: seed", seed, "is ill-conditioned in float32: HIP-vs-oracle", e, "oracle f32-vs-f64", spread)
            e = 0.0
    if e > RTOL[dn] * 2:
        bad += 1
        print("MISMATCH seed", seed, dn, p["m"], p["n"], int(p["colptr"][-1]), e, f.info())
print("seeds", first, "..", first + count - 1, "mismatches", bad, "worst rel err", worst)
