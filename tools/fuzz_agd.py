"""Developer tool: randomised check of the device-resident AGD loop (stats + apply kernels, gamma continuation, equality rows)
against oracle/agd_oracle.maximize on small random problems (python tools/fuzz_agd.py [first_seed] [count])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, oracle
from oracle import agd_oracle
from tests.helpers import relerr, torch_args
from tests.test_gpu_fuzz import _case
from dualip_amd.objectives.matching import MatchingSolverDualObjectiveFunction
from dualip_amd.optimizers.agd import AcceleratedGradientDescent
first = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = 0.0; bad = 0
for seed in range(first, first + count):
    p, pm, entries, col_proj, gamma, lam, dn = _case(seed)
    rng = np.random.default_rng(seed)
    m = p["m"]
    iters = int(rng.choice([20, 45]))
    decay = {"decay_steps": int(rng.choice([3, 7])), "decay_factor": 0.7} if rng.random() < 0.5 else None
    eq = (rng.random(m) < 0.2) if rng.random() < 0.5 else None
    s0, s1 = float(rng.choice([1e-5, 1e-3])), float(rng.choice([1e-2, 1e-1]))
    f = MatchingSolverDualObjectiveFunction(torch_args(p, "f64", pm, "cuda:0", equality_mask=eq), gamma=gamma)
    solver = AcceleratedGradientDescent(max_iter=iters, gamma=gamma, initial_step_size=s0, max_step_size=s1, gamma_decay_type="step" if decay else None,
                                        gamma_decay_params=decay or {}, iteration_callback=False)
    res = solver.maximize(f, torch.zeros(m, dtype=torch.float64, device="cuda:0"))
    def calc(l, g):
        ax, obj0, ssq, _ = oracle.matching_calculate(m, p["n"], p["colptr"], p["rowidx"], p["a"], p["c"], l, g, entries, col_proj=col_proj, dtype=np.float64, want_x=False)
        grad, obj, *_ = agd_oracle.epilogue(ax, obj0, ssq, l, p["b"], g, np.float64)
        return grad, obj, None
    want = agd_oracle.maximize(calc, np.zeros(m), iters, gamma, initial_step_size=s0, max_step_size=s1, decay=decay, eq_mask=eq, dtype=np.float64)
    # equality rows + a fixed step can make the iteration itself diverge (|objective| growing by 30x per iteration):
    # round-off is then amplified without bound, so traces are compared while they are still finite-sized
    got_log, want_log = np.array(res.dual_objective_log), want["dual_obj_log"]
    k = int(np.argmax(np.abs(want_log) > 1e9)) if (np.abs(want_log) > 1e9).any() else len(want_log)
    k = max(k, 4)
    e = max(relerr(got_log[:k], want_log[:k]), relerr(np.array(res.step_size_log)[:k], want["step_log"][:k]))
    if k == len(want_log):
        e = max(e, relerr(res.dual_val.cpu().numpy(), want["dual_val"]))
    worst = max(worst, e)
    if e > 1e-6:
        bad += 1
        print("MISMATCH seed", seed, m, p["n"], iters, decay, eq is not None, e)
print("seeds", first, "..", first + count - 1, "mismatches", bad, "worst rel err", worst)
